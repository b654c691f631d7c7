#!/usr/bin/env python
"""Benchmark of the session-sequence hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU, RCCL over xGMI.  Either launched by `python -m torch.distributed.run
   --nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or called
   plainly as `python bench.py --gpus N ...`: with WORLD_SIZE unset the script re-executes itself under
   torch.distributed.run on 127.0.0.1 and passes rank 0's JSON line through -- the reference's launch contract,
   docs/source/multi_gpu_train.md:27-40)

A "step" = one full training pass of the hot path over one synthetic batch already resident in
HBM: masking -> embedding gather -> 4-layer XLNet -> next-item head (tied full softmax) ->
backward -> gradient exchange -> fused Adam.
Workload = BASELINE.json configs[1]: item vocab 100k, d_model 128, 4 layers, 4 heads, seq 20,
per-GPU batch 1024 (weak scaling: global batch 1024*N; 8192 at N=8), MLM p=0.15, fp32, every
reference dropout site active (p = 0.3).
Prints ONE JSON line on rank 0: the metric/value contract + `roofline` (+ `roofline_gather`) +
`cpu_baseline` + `recall_at_20` (the other half of BASELINE.json's metric).

Data-parallel wiring (N > 1): the table bucket's dense part (the tied head's d W) is final right after
the head's backward: its all-reduce is launched there, on its own stream, under the transformer's
backward; the lookup scatter of the input block travels row-sparse ((ids, rows) all-gather + the
deterministic sorted scatter on every rank); the small dense bucket is reduced at the end.
The functions below are importable without a GPU: tests/test_distributed_cpu.py drives
`make_train_step` / `timed_region` with a stub model over gloo.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V_ITEMS, D_MODEL, N_LAYER, N_HEAD, SEQ, BATCH = 100_000, 128, 4, 4, 20, 1024
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 / _f16, dense (no sparsity)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def rank_seeds(rank):
    """per-rank Philox keys of the MLM draws and the dropout masks (ranks must not replay each other)"""
    return 1234 + rank, 4321 + rank


# BASELINE.json configs[2] (C3) = configs[1] + three categoricals + two continuous SoftEmbedding features, concat merge,
# ReLU projection 336 -> d_model (reference builder: transformers4rec/torch/features/sequence.py:140-156).  Cardinalities are
# this repo's choice: the reference's synthetic recipe takes them from the schema.
C3_CATS, C3_CONTS = (("category", 1000), ("brand", 100), ("dow", 10)), ("price", "age")
WORKLOADS = {
    "c2": "BASELINE.json configs[1]: synthetic schema, item vocab 100k (100001 table rows), d_model 128, 4-layer 4-head XLNet, "
          "seq_len 20, per-GPU batch 1024, MLM p=0.15, tied-weight full softmax, Adam, fwd+bwd+gradient exchange+optimizer per step",
    "c3": "BASELINE.json configs[2]: configs[1]'s 100k-item XLNet with MULTI-FEATURE input -- item-id (128) + 3 categoricals "
          "(cardinality 1000 / 100 / 10, 64 wide) + 2 continuous SoftEmbedding features (8 wide), concat merge (336) -> ReLU "
          "projection to d_model 128 -- seq_len 20, per-GPU batch 1024, MLM p=0.15, tied-weight full softmax, Adam, "
          "fwd+bwd+gradient exchange+optimizer per step (the configuration BASELINE names for the 8-GPU DP run)",
}


def build_modules(dropout, v_items=V_ITEMS, d_model=D_MODEL, n_layer=N_LAYER, n_head=N_HEAD, seq=SEQ, config="c2", seed=0):
    """the benchmarked model on the CPU, freshly initialised from torch.manual_seed(seed) (no device, no optimizer): what
    `build` moves to the GPU, and what oracle/cpu_lockstep.py takes its identical initial parameters from"""
    import transformers4rec_amd as tr

    torch.manual_seed(seed)
    if config == "c3":
        schema = tr.session_schema(v_items, seq, C3_CATS, C3_CONTS)
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=seq, masking="mlm", aggregation="concat",
                                                        d_output=d_model, continuous_soft_embeddings=True,
                                                        embedding_dims={"item_id": d_model}, embedding_dim_default=64)
    else:
        schema = tr.session_schema(v_items, seq)
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=seq, masking="mlm",
                                                        embedding_dim_default=d_model)
    cfg = tr.XLNetConfig.build(d_model, n_head, n_layer, total_seq_length=seq, dropout=dropout)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
    return tr, schema, model


def build(device, dropout, v_items=V_ITEMS, d_model=D_MODEL, n_layer=N_LAYER, n_head=N_HEAD, seq=SEQ, lr=1e-3, config="c2", seed=0):
    tr, schema, model = build_modules(dropout, v_items, d_model, n_layer, n_head, seq, config, seed)
    model.to(device)
    dense, tables = tr.flatten_model(model)
    opt = tr.FusedAdam([dense, tables], lr=lr)
    return tr, schema, model, dense, tables, opt


def table_exchange_mode():
    """how the embedding-table gradient crosses the ranks (T4R_BENCH_TABLE_EXCHANGE):
      "sparse" (default)  the tied head's dense d W is all-reduced EARLY (right after the head's backward, under the body's
                          backward); the lookup scatter travels as (ids, rows): all-gather + the deterministic sorted scatter
                          on every rank (84 MB gathered per step at C2 / 8 GPUs, exposed);
      "dense"             every rank scatters its own lookups into its table gradient, then ONE table all-reduce (51 MB) at the
                          end of the backward, exposed, nothing row-sparse.
    Both give the same gradients (tests/test_distributed_cpu.py); which one is faster is a property of the fabric: measured by
    the driver's 8-GPU run (`config.table_exchange` in the JSON line says which was used)."""
    m = os.environ.get("T4R_BENCH_TABLE_EXCHANGE", "auto")
    if m not in ("sparse", "dense", "auto"):
        raise SystemExit("T4R_BENCH_TABLE_EXCHANGE must be sparse, dense or auto")
    return m


def teardown_data_parallel(tables, hook):
    """undo setup_data_parallel: the head-backward hook and the row-sparse sinks of the table parameters"""
    import transformers4rec_amd as tr

    if hook is not None:
        hook.remove()
    if tables is not None:
        tr.SparseRowExchange.detach(*[p for _, p, _ in tables.entries])


def pick_table_exchange(setup, teardown, make_step, world, device, steps=3, warm=2):
    """"auto" (the default at N > 1): time `steps` training steps of each form of the table-gradient exchange on THIS
    fabric -- barrier + device sync on both sides, MAX over ranks, so every rank takes the same decision -- and keep the
    faster.  setup(mode) -> (reducer, hook); teardown(hook); make_step(reducer) -> train_step.
    -> (mode, reducer, hook, {"sparse": ms per step, "dense": ms per step})"""
    import torch.distributed as dist

    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    times = {}
    for mode in ("sparse", "dense"):
        reducer, hook = setup(mode)
        step = make_step(reducer)
        for i in range(warm):
            step(i)
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warm + i)
        if world > 1:
            dist.barrier()
        sync()
        t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times[mode] = 1e3 * float(t.item()) / steps
        teardown(hook)
    best = min(times, key=times.get)        # identical on every rank: the times were all-reduced
    reducer, hook = setup(best)
    return best, reducer, hook, {k: round(v, 4) for k, v in times.items()}


def setup_data_parallel(tr, model, dense, tables, world, mode=None):
    """-> (reducer, hook handle or None).  N > 1: async table all-reduce after the head's backward + row-sparse
    exchange of the lookup scatter (mode "sparse"), or local scatter + one late table all-reduce (mode "dense");
    N = 1: everything local (the reducer only runs the local scatter)."""
    sparse = None
    hook = None
    mode = mode or table_exchange_mode()
    if mode == "auto":
        mode = "sparse"         # the caller runs pick_table_exchange; a plain call keeps the row-sparse form
    if mode == "dense" and os.environ.get("T4R_BENCH_SPARSE", "0") != "1":
        return tr.GradReducer(dense.grad, tables.grad if tables is not None else None, sparse=None), None
    if world > 1 or os.environ.get("T4R_BENCH_SPARSE", "0") == "1":
        # full-softmax head: only the lookup scatter is row-sparse, B * L rows on every rank (weak scaling)
        sparse = tr.SparseRowExchange(equal_sizes=True)
        sparse.attach(*[p for _, p, _ in tables.entries])
    reducer = tr.GradReducer(dense.grad, tables.grad if tables is not None else None, sparse=sparse)
    if world > 1 and sparse is not None:
        hook = tr.head_backward_hook(model, reducer.reduce_tables_async)
    return reducer, hook


def make_train_step(model, batches, reducer, opt):
    def train_step(i):
        x = batches[i % len(batches)]
        out = model(x, training=True)
        out["loss"].backward()          # N > 1: the head-backward hook launches the table all-reduce here
        reducer.reduce_all()
        opt.step(grad_scale=reducer.grad_scale)
        return out

    return train_step


def _lib_int(name):
    from transformers4rec_amd import _lib

    return int(getattr(_lib.load(), name)())


def timed_region(train_step, warmup, steps, world, device, first_step=0):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync, MAX over ranks.
    -> (seconds, last output, label rows seen)"""
    import torch.distributed as dist

    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    out = None
    for i in range(warmup):
        out = train_step(first_step + i)
    barrier()
    t0 = time.perf_counter()
    n_lab = 0
    for i in range(steps):
        out = train_step(first_step + warmup + i)
        n_lab += out["labels"].numel()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item()), out, n_lab


def extra_windows(train_step, steps, world, device, first_step, n_windows=5):
    """n_windows MORE timed windows of `steps` steps each, after the contract's window (training simply continues): the
    spread of the headline between windows of one process -- min / median / max ms per step, MAX over ranks each."""
    import torch.distributed as dist

    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    ms = []
    for wdx in range(n_windows):
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            train_step(first_step + wdx * steps + i)
        if world > 1:
            dist.barrier()
        sync()
        t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms.append(1e3 * float(t.item()) / steps)
    srt = sorted(ms)
    return {"n_windows": n_windows, "steps_per_window": steps, "min": round(srt[0], 4), "median": round(srt[len(srt) // 2], 4),
            "max": round(srt[-1], 4), "all": [round(x, 4) for x in ms]}


# --------------------------------------------------------------------------------------------- CPU leg
def cpu_baseline_c3(cores, seconds_budget, max_steps):
    """configs[2] on the oracle: the multi-feature session forward (oracle/t4r_oracle.py session_forward: gathers, soft
    embeddings + LayerNorm, concat, ReLU projection, XLNet, tied head) + backward + Adam at full size.  Dropout 0 here (the
    oracle's multi-feature entry has no dropout masks): the CPU figure is, if anything, flattered."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_utils as gu
    import t4r_oracle as O
    import transformers4rec_amd as tr

    _, schema, model, _, _, _ = build(torch.device("cpu"), 0.0, config="c3")
    p = gu.oracle_params({"p/" + k: v.detach().clone().numpy() for k, v in model.state_dict().items()}, requires_grad=True)
    leaves, stack = [], [p]
    while stack:            # every trainable tensor of the oracle's parameter tree
        o = stack.pop()
        if isinstance(o, torch.Tensor):
            if o.requires_grad:
                leaves.append(o)
        elif isinstance(o, dict):
            stack.extend(o.values())
        elif isinstance(o, (list, tuple)):
            stack.extend(o)
    opt = torch.optim.Adam(leaves, lr=1e-3)
    cfg = dict(n_head=N_HEAD, eps=0.03, item="item_id", masking="mlm")

    def step(seed):
        data = tr.random_data_from_schema(schema, BATCH, SEQ, seed=seed)
        ids = data["item_id"]
        bern = torch.rand(BATCH, SEQ) < 0.15
        j1 = (torch.rand(BATCH) * (ids != 0).sum(1)).long()
        m, lab = O.mlm_targets_train(ids, bern, j1, lambda mm: mm.float().argmax(1))
        opt.zero_grad()
        O.session_forward(p, cfg, data, m, lab, True, False)["loss"].backward()
        opt.step()

    step(0)
    t0 = time.perf_counter()
    n_steps = 0
    while True:
        step(1 + n_steps)
        n_steps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n_steps >= max_steps:
            break
    return {"value": round(BATCH * n_steps / el, 2), "unit": "sessions/s", "cores": cores, "kind": "port",
            "sample": f"{n_steps} train steps (fwd+bwd+Adam) of batch {BATCH} of configs[2] at full size (V=100001, d=128, 4 layers, seq 20, "
                      f"item + 3 categoricals + 2 soft embeddings, concat 336 -> 128), dropout 0, oracle/t4r_oracle.py session_forward on "
                      f"{cores} host threads"}


def cpu_baseline(dropout, seconds_budget=25.0, max_steps=8, config="c2"):
    """The oracle ("port" of the reference algorithm, plain torch fp32 on the host cores) timed on a bounded
    sample of the SAME workload: full V / d_model / layers / batch 1024 and the reference's dropout law
    (one torch.bernoulli mask per dropout site per step, as nn.Dropout draws them on the CPU path)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import t4r_oracle as O
    import transformers4rec_amd as tr

    # torch CPU ops stop scaling (and oversubscribe badly) far below the 256 hardware threads of
    # the MI355X host: 32 threads measured best on the GPU box; the count used is reported.
    cores = min(os.cpu_count() or 1, int(os.environ.get("T4R_CPU_BASELINE_THREADS", "32")))
    torch.set_num_threads(cores)
    if config == "c3":
        return cpu_baseline_c3(cores, seconds_budget, max_steps)
    B = BATCH
    g = torch.Generator().manual_seed(0)
    rn = lambda *s, std=0.01: (std * torch.randn(*s, generator=g)).requires_grad_()
    D, n, dh = D_MODEL, N_HEAD, D_MODEL // N_HEAD
    layers = [dict(q=rn(D, n, dh), k=rn(D, n, dh), v=rn(D, n, dh), o=rn(D, n, dh), r=rn(D, n, dh),
                   r_w_bias=rn(n, dh), r_r_bias=rn(n, dh), ln_w=torch.ones(D, requires_grad=True),
                   ln_b=torch.zeros(D, requires_grad=True), w1=rn(4 * D, D), b1=torch.zeros(4 * D, requires_grad=True),
                   w2=rn(D, 4 * D), b2=torch.zeros(D, requires_grad=True),
                   ff_ln_w=torch.ones(D, requires_grad=True), ff_ln_b=torch.zeros(D, requires_grad=True))
              for _ in range(N_LAYER)]
    table, memb = rn(V_ITEMS + 1, D, std=0.05), rn(D, std=0.001)
    leaves = [table, memb] + [t for lp in layers for t in lp.values()]
    opt = torch.optim.Adam(leaves, lr=1e-3)
    schema = tr.session_schema(V_ITEMS, SEQ)
    p = dropout
    keep = lambda *shape: torch.bernoulli(torch.full(shape, 1.0 - p))

    def step(seed):
        ids = tr.random_data_from_schema(schema, B, SEQ, seed=seed)["item_id"]
        bern = torch.rand(B, SEQ) < 0.15
        lens = (ids != 0).sum(1)
        j1 = (torch.rand(B) * lens).long()
        m, lab = O.mlm_targets_train(ids, bern, j1, lambda mm: mm.float().argmax(1))
        opt.zero_grad()
        x = O.apply_mask_mlm(O.embedding_lookup(ids, table), m, memb, True, False)
        if p > 0:
            s = 1.0 / (1.0 - p)
            h = x * keep(B, SEQ, D) * s                                   # HF :1116
            pos_mask = keep(B, 2 * SEQ, D)                                # HF :1143, once per forward
            for lp in layers:
                masks = dict(pos=pos_mask, prob=keep(B, n, SEQ, SEQ), attn_out=keep(B, SEQ, D),
                             ff_act=keep(B, SEQ, 4 * D), ff_out=keep(B, SEQ, D))
                h = O.xlnet_layer_dropout(h, lp, n, 0.03, masks, p)
            h = h * keep(B, SEQ, D) * s                                   # HF :1177
        else:
            h = O.xlnet_model(x, layers, n, 0.03)
        xr, y = O.remove_pad_rows(h, lab)
        loss = O.cross_entropy(O.head_logits(xr, table, 1.0), y)
        loss.backward()
        opt.step()

    step(0)  # warm-up (page-faults ~1 GB of logits)
    t0 = time.perf_counter()
    n_steps = 0
    while True:
        step(1 + n_steps)
        n_steps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n_steps >= max_steps:
            break
    return {"value": round(B * n_steps / el, 2), "unit": "sessions/s", "cores": cores, "kind": "port",
            "sample": f"{n_steps} train steps (fwd+bwd+Adam) of batch {B} at full V=100001, d=128, 4 layers, "
                      f"seq 20, dropout {p} (a torch.bernoulli mask per site), oracle/t4r_oracle.py on {cores} "
                      "host threads; the reference-verbatim CPU record (build container) is "
                      "`reference_verbatim` beside this"}


def cpu_baseline_reference(dropout, steps=3, threads=None):
    """The UNMODIFIED reference (transformers4rec.torch + HF XLNet, imported from /root/reference through
    oracle/ref_standins.py) timed on THIS host's cores on the same workload -- only where the reference tree exists
    (the build container; the GPU box has no /root/reference, there the committed measurement is quoted instead).
    -> dict or None"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import ref_standins as rs
    except Exception:      # noqa: BLE001
        return None
    if not os.path.isdir(rs.REFERENCE_ROOT):
        for name in ("r06_cpu_reference_bench.json", "r02_cpu_reference_bench.json"):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            with open(path) as f:
                j = json.load(f)
            n = j["steps_timed"]
            return {"value": j["sessions_per_s_median"], "unit": "sessions/s", "cores": j["threads"], "kind": "reference",
                    "measured_here": False, "median": j["sessions_per_s_median"], "best": j["sessions_per_s_best"],
                    "worst": j.get("sessions_per_s_worst"), "n_steps": n, "date": j.get("date"),
                    "sample": f"committed measurement profiles/{name} (build container, {j['threads']} threads of a microVM whose "
                              "identical runs have differed several-fold between days; /root/reference does not exist on this "
                              f"box): {j['what']}"}
        return None
    import make_golden as mg
    import transformers4rec_amd as hip

    tr = rs.import_reference()
    from transformers4rec.config import transformer as tconf

    cores = threads or min(os.cpu_count() or 1, int(os.environ.get("T4R_CPU_BASELINE_THREADS", "32")))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    inputs = tr.TabularSequenceFeatures.from_schema(mg.make_schema(V_ITEMS, SEQ), max_sequence_length=SEQ, masking="mlm",
                                                    embedding_dim_default=D_MODEL)
    cfg = tconf.XLNetConfig.build(d_model=D_MODEL, n_head=N_HEAD, n_layer=N_LAYER, total_seq_length=SEQ, dropout=dropout)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    hschema = hip.session_schema(V_ITEMS, SEQ)
    times = []
    for i in range(1 + steps):
        x = hip.random_data_from_schema(hschema, BATCH, SEQ, seed=100 + i)
        t0 = time.perf_counter()
        opt.zero_grad()
        out = model(x, training=True)
        out["loss"].backward()
        opt.step()
        if i:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(BATCH / med, 2), "unit": "sessions/s", "cores": cores, "kind": "reference", "measured_here": True,
            "median": round(BATCH / med, 2), "best": round(BATCH / times[0], 2), "worst": round(BATCH / times[-1], 2),
            "n_steps": len(times), "date": time.strftime("%Y-%m-%d"),
            "sample": f"{steps} train steps (fwd+bwd+Adam) of batch {BATCH}, unmodified transformers4rec.torch + HF XLNetModel "
                      f"on {cores} host threads of this box, median"}


# --------------------------------------------------------------------------------------------- Recall@20
MARKOV_FANOUT, MARKOV_P_FOLLOW = 2, 0.9
RECALL_TRAIN_STEPS = 200


def markov_sessions(n, seq, active, seed, p_follow=MARKOV_P_FOLLOW, min_len=5, fanout=MARKOV_FANOUT):
    """sessions of a fixed first-order Markov chain over the item ids `active`: with probability p_follow the next item
    is one of `fanout` fixed successors of the current one, drawn with Zipf(1) weights (1 / rank), else uniform; lengths
    ~ U[min_len, seq] as the reference's synthetic recipe (torch/utils/schema_utils.py:72-80).  Learnable signal for
    Recall@20 / NDCG@20.  Round 4's single-successor chain put every implementation AT the generator's ceiling
    (Recall@20 = p_follow to four digits, HIP and CPU oracle alike); with weighted successors the model has to RANK them:
    NDCG@20 -- the quality headline -- has its ceiling at the exact successor order (`markov_bayes`), and after 200 steps
    both figures are still moving, so a regression in the kernels shows up in them.  Two successors with weights 1 : 1/2 is the
    chain tools/recall_sweep.py picked (HIP path, benchmarked configuration, 200 steps): Recall@20 0.76 of a 0.90 ceiling, NDCG@20
    0.50 of 0.79 -- one successor sits at the ceiling after 200 steps, four or more are still on the initial loss plateau."""
    g = torch.Generator().manual_seed(seed)
    gs = torch.Generator().manual_seed(12345)                    # the chain itself is fixed
    A = active.numel()
    succ = torch.stack([torch.randperm(A, generator=gs) for _ in range(fanout)], 1)      # [A, fanout]
    w = 1.0 / torch.arange(1, fanout + 1, dtype=torch.float64)
    cur = torch.randint(0, A, (n,), generator=g)
    cols = [cur]
    for _ in range(seq - 1):
        follow = torch.rand(n, generator=g) < p_follow
        k = torch.multinomial(w, n, replacement=True, generator=g)
        cur = torch.where(follow, succ[cur, k], torch.randint(0, A, (n,), generator=g))
        cols.append(cur)
    idx = torch.stack(cols, 1)
    lens = torch.randint(min_len, seq + 1, (n,), generator=g)
    m = torch.arange(seq)[None] < lens[:, None]
    return active[idx] * m


def session_features(ids, config="c2"):
    """the input dict of a batch of item-id sessions: C2 = the ids; C3 = + categoricals / continuous features that are fixed
    functions of the item (category, brand, day slot, price, age: item metadata, as in a real catalogue), 0 at padding"""
    x = {"item_id": ids}
    if config == "c3":
        live = ids != 0
        for name, card in C3_CATS:
            x[name] = (1 + (ids * 7919 + len(name)) % (card - 1)) * live
        for j, name in enumerate(C3_CONTS):
            x[name] = (((ids * (2654435761 + 40503 * j)) % 1000).float() / 1000.0) * live
    return x


def markov_bayes(k=20, n_active=2000, p_follow=MARKOV_P_FOLLOW, fanout=MARKOV_FANOUT):
    """(Recall@k, NDCG@k) of the predictor that knows the chain: the ceilings of `markov_sessions` (successor collisions and
    the uniform part's ranks beyond the successors ignored)"""
    import math

    w = [1.0 / i for i in range(1, fanout + 1)]
    tot = sum(w)
    top = min(k, fanout)
    rec = p_follow * sum(w[:top]) / tot + (1.0 - p_follow) * k / n_active
    ndcg = p_follow * sum(w[i] / math.log2(i + 2) for i in range(top)) / tot
    return rec, ndcg


PROBE_KEYS = (0, 1234, 4321)          # (init seed, MLM Philox key, dropout Philox key) of the lockstep trajectory
PROBE_SEEDS = 8                       # independent runs of the same procedure for the spread


def probe_keys(s):
    """the (init seed, MLM key, dropout key) of independent probe run s (oracle/cpu_lockstep.py replays 0..2 on the CPU)"""
    return 100 + s, 5000 + 17 * s, 9000 + 31 * s


def train_probe(device, dropout, steps, keys, config="c2"):
    """`steps` training steps of the benchmarked configuration (Adam lr 2e-3) on bench.markov_sessions from the given
    (init seed, MLM key, dropout key), the loss of every step, then Recall@20 / NDCG@20 on four held-out batches through the
    fused evaluation head (ranks inside the logits GEMM).  Bit-reproducible: same keys, same trajectory."""
    init_seed, mask_seed, drop_seed = keys
    tr, schema, model, dense, tables, opt = build(device, dropout, lr=2e-3, config=config, seed=init_seed)
    table = model.input_features.item_embedding_table.weight
    q0 = model.transformer_block.transformer.layer[0].rel_attn.q
    checksum = [round(float(table.detach().double().abs().sum()), 6), round(float(q0.detach().double().abs().sum()), 9)]
    model.input_features.masking.seed = mask_seed
    model.transformer_block.transformer.seed = drop_seed
    active = 1 + torch.arange(2000) * (V_ITEMS // 2000)
    to_dev = lambda d: {k: v.to(device) for k, v in d.items()}
    model.train()
    losses = []
    t0 = time.perf_counter()
    for i in range(steps):
        x = to_dev(session_features(markov_sessions(BATCH, SEQ, active, 10 + i), config))
        out = model(x, training=True)
        out["loss"].backward()
        opt.step()
        losses.append(out["loss"].detach())
    losses = [float(v) for v in torch.stack(losses).cpu()]
    model.eval()
    task = model.prediction_task
    task.reset_metrics()
    with torch.no_grad():
        for j in range(4):
            x = to_dev(session_features(markov_sessions(BATCH, SEQ, active, 900_000 + j), config))
            task.evaluate_ranks(model.heads[0].body(x, training=False, testing=True))
    mt = task.compute_metrics()
    torch.cuda.synchronize()
    return {"init_checksum": checksum, "loss_per_step": [round(v, 6) for v in losses],
            "recall_at_20": round(mt["next-item/recall_at_20"], 4), "ndcg_at_20": round(mt["next-item/ndcg_at_20"], 4),
            "avg_precision_at_20": round(mt["next-item/avg_precision_at_20"], 4),
            "final_train_loss": round(losses[-1], 4), "train_steps": steps, "seconds": round(time.perf_counter() - t0, 2)}


def spread(rows, key):
    import statistics

    v = [r[key] for r in rows]
    return {"mean": round(statistics.mean(v), 4), "sd": round(statistics.stdev(v), 4) if len(v) > 1 else None,
            "min": min(v), "max": max(v), "n": len(v)}


def _committed(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def recall_probe(device, dropout, train_steps=RECALL_TRAIN_STEPS, lockstep_steps=600, config="c2", n_seeds=PROBE_SEEDS):
    """Recall@20 / NDCG@20 of next-item prediction on a held-out split after K training steps on Markov-chain sessions.
    (a) the benchmarked configuration on the HIP path from fixed keys -- the trajectory oracle/cpu_lockstep.py replays on the
        CPU oracle with the SAME decisions at every random site (oracle/device_rng.py restates the device's Philox streams;
        tests/test_round6_gpu.py checks that bit for bit): this run's per-step losses are compared with the committed CPU
        replay (profiles/r06_lockstep_cpu.json), so every bench line carries a live whole-trajectory parity figure at
        dropout 0.3;
    (a') `n_seeds` independent runs of the same procedure (other init / keys): mean, sd, min, max -- the spread two
        independent samples of this procedure show (round 5 compared ONE HIP sample with ONE CPU sample and could not say
        whether 0.764 vs 0.792 was a bug: it is the spread) -- beside the committed CPU replays of the first three;
    (b) a reduced configuration trained in lockstep with the CPU oracle run live in this process (dropout 0)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_utils as gu
    import t4r_oracle as O

    res = {}
    res["generator"] = {"what": f"first-order Markov chain over 2000 active items, {MARKOV_FANOUT} Zipf-weighted successors per item, "
                                f"p_follow {MARKOV_P_FOLLOW} (bench.markov_sessions)",
                        "bayes_optimal_recall_at_20": round(markov_bayes()[0], 4), "bayes_optimal_ndcg_at_20": round(markov_bayes()[1], 4),
                        "note": "the successors must be ranked: NDCG@20 is the quality headline; both figures still move after 200 "
                                "steps (the loss leaves its plateau around step 120), so independent runs differ visibly"}
    # ---- (a) the lockstep trajectory
    hip = train_probe(device, dropout, train_steps, PROBE_KEYS, config)
    res["hip_bench_config"] = {k: hip[k] for k in ("recall_at_20", "ndcg_at_20", "avg_precision_at_20", "train_steps",
                                                   "final_train_loss", "seconds")}
    res["hip_bench_config"]["eval_sessions"] = 4 * BATCH
    res["hip_bench_config"]["keys"] = "init seed %d, MLM key %d, dropout key %d" % PROBE_KEYS
    cpu = _committed("r06_lockstep_cpu.json") if (config == "c2" and abs(dropout - 0.3) < 1e-9 and train_steps == RECALL_TRAIN_STEPS) else None
    if cpu is not None and cpu.get("init_checksum") == hip["init_checksum"]:
        d = [abs(a - b) for a, b in zip(hip["loss_per_step"], cpu["loss_per_step"])]
        res["lockstep_bench_config"] = {
            "what": "this run's HIP trajectory vs the committed CPU-oracle replay of the same trajectory (same init, sessions, MLM "
                    "targets and dropout masks at every site of every step: oracle/cpu_lockstep.py + oracle/device_rng.py)",
            "max_abs_loss_diff_first_50_steps": round(max(d[:50]), 7), "max_abs_loss_diff_all": round(max(d), 5),
            "first_step_with_loss_diff_above_1e-3": next((i for i, x in enumerate(d) if x > 1e-3), None),
            "hip": {k: hip[k] for k in ("recall_at_20", "ndcg_at_20", "final_train_loss")},
            "cpu_oracle": {k: cpu[k] for k in ("recall_at_20", "ndcg_at_20", "final_train_loss")},
            "cpu_source": "committed: profiles/r06_lockstep_cpu.json (%s s on %s host threads of the build container)"
                          % (cpu.get("train_seconds"), cpu.get("host_threads"))}
    # ---- (a') spread over independent runs
    if n_seeds:
        rows = [train_probe(device, dropout, train_steps, probe_keys(s), config) for s in range(n_seeds)]
        sp = {"what": f"{n_seeds} independent runs (init seed 100+s, MLM key 5000+17s, dropout key 9000+31s), {train_steps} steps each",
              "hip": {k: spread(rows, k) for k in ("recall_at_20", "ndcg_at_20", "final_train_loss")}}
        cpu_rows = []
        for s_ in range(n_seeds):
            cj = _committed(f"r06_lockstep_cpu_seed{s_}.json") if config == "c2" else None
            if cj is not None and cj.get("init_checksum") == rows[s_]["init_checksum"]:
                cpu_rows.append((s_, cj))
        if cpu_rows:
            sp["cpu_oracle_committed"] = {k: spread([cj for _, cj in cpu_rows], k) for k in ("recall_at_20", "ndcg_at_20", "final_train_loss")}
            sp["same_keys_pairs"] = [{"run": s_, "hip": {k: rows[s_][k] for k in ("recall_at_20", "ndcg_at_20", "final_train_loss")},
                                      "cpu_oracle": {k: cj[k] for k in ("recall_at_20", "ndcg_at_20", "final_train_loss")},
                                      "max_abs_loss_diff_first_50_steps": round(max(abs(a - b) for a, b in zip(
                                          rows[s_]["loss_per_step"][:50], cj["loss_per_step"][:50])), 7)} for s_, cj in cpu_rows]
        res["seed_spread"] = sp
    # ---- (b) lockstep HIP / oracle at reduced size
    Vr, Br, Dr, NLr = 2000, 256, 64, 2
    tr, schema, model, dense, tables, opt = build(device, 0.0, v_items=Vr, d_model=Dr, n_layer=NLr, n_head=4, lr=5e-3)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    p = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
    leaves = [p["tables"]["item_id"], p["masked_item_embedding"]] + [t for lp in p["layers"] for t in lp.values()]
    oopt = torch.optim.Adam(leaves, lr=5e-3)
    cfg = dict(n_head=4, eps=0.03, item="item_id", masking="mlm")
    active_r = 1 + torch.arange(Vr - 1)
    masking = model.input_features.masking
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    model.train()
    for i in range(lockstep_steps):
        ids = markov_sessions(Br, SEQ, active_r, 50_000 + i)
        out = model({"item_id": ids.to(device)}, training=True)
        out["loss"].backward()
        opt.step()
        oopt.zero_grad()
        ref = O.session_forward(p, cfg, {"item_id": ids}, masking.mask_schema.cpu(), masking.masked_targets.cpu(),
                                True, False)
        ref["loss"].backward()
        oopt.step()
    model.eval()
    task = model.prediction_task
    task.reset_metrics()
    rec_o, ndcg_o, n_o = 0.0, 0.0, 0
    with torch.no_grad():
        for j in range(2):
            ids = markov_sessions(512, SEQ, active_r, 990_000 + j)
            h = model.heads[0].body({"item_id": ids.to(device)}, training=False, testing=True)
            task.evaluate_ranks(h)
            m, lab = O.mlm_targets_eval(ids)
            ro = O.session_forward(p, cfg, {"item_id": ids}, m, lab, False, True)
            rec_o += float(O.recall_at_k(ro["logits"], ro["labels"], 20).sum())
            ndcg_o += float(O.ndcg_at_k(ro["logits"], ro["labels"], 20).sum())
            n_o += ro["labels"].numel()
    mt = task.compute_metrics()
    res["lockstep_reduced"] = {
        "config": f"V={Vr}, d={Dr}, {NLr} layers, batch {Br}, {lockstep_steps} steps, dropout 0, same init / masks / Adam",
        "hip": {"recall_at_20": round(mt["next-item/recall_at_20"], 4), "ndcg_at_20": round(mt["next-item/ndcg_at_20"], 4),
                "final_train_loss": round(float(out["loss"].detach()), 5)},
        "cpu_oracle": {"recall_at_20": round(rec_o / n_o, 4), "ndcg_at_20": round(ndcg_o / n_o, 4),
                       "final_train_loss": round(float(ref["loss"].detach()), 5)}}
    return res


def recall_probe_dp(device, dropout, world, rank, train_steps=RECALL_TRAIN_STEPS, config="c2"):
    """N > 1 form of part (a) of `recall_probe`: the benchmarked configuration trained data-parallel (every rank its own
    Markov sessions, the same gradient exchange as the timed steps), evaluated on held-out sessions SHARDED over the
    ranks; `compute_metrics()` all-reduces the (sum, count) state, so the value is the mean over every rank's label rows
    (the reference cat-syncs its torchmetrics state: ranking_metric.py:50, trainer.py:519-525).  Collective: every rank
    runs it."""
    tr, schema, model, dense, tables, opt = build(device, dropout, lr=2e-3, config=config)
    to_dev = lambda d: {k: v.to(device) for k, v in d.items()}
    masking = model.input_features.masking
    masking.seed, model.transformer_block.transformer.seed = rank_seeds(rank)
    reducer, _hook = setup_data_parallel(tr, model, dense, tables, world)
    active = 1 + torch.arange(2000) * (V_ITEMS // 2000)
    model.train()
    t0 = time.perf_counter()
    for i in range(train_steps):
        x = to_dev(session_features(markov_sessions(BATCH, SEQ, active, 10 + i * world + rank), config))
        out = model(x, training=True)
        out["loss"].backward()
        reducer.reduce_all()
        opt.step(grad_scale=reducer.grad_scale)
    model.eval()
    task = model.prediction_task
    task.reset_metrics()
    with torch.no_grad():
        for j in range(4):
            x = to_dev(session_features(markov_sessions(BATCH, SEQ, active, 900_000 + j * world + rank), config))
            task.evaluate_ranks(model.heads[0].body(x, training=False, testing=True))
    mt = task.compute_metrics()             # all-reduce over the ranks
    torch.cuda.synchronize()
    return {"hip_bench_config_dp": {
        "recall_at_20": round(mt["next-item/recall_at_20"], 4), "ndcg_at_20": round(mt["next-item/ndcg_at_20"], 4),
        "avg_precision_at_20": round(mt["next-item/avg_precision_at_20"], 4), "train_steps": train_steps,
        "global_batch": BATCH * world, "eval_sessions": 4 * BATCH * world,
        "final_train_loss_rank0": round(float(out["loss"].detach()), 4), "seconds": round(time.perf_counter() - t0, 2),
        "note": "metric state (sum, count) all-reduced over the ranks in compute_metrics"}}


# --------------------------------------------------------------------------------------------- 8-rank exchange, rehearsed on one GPU
XGMI_LINK_GBS = 153.0       # per direction and link, 7 links per GPU (task statement / MI355X_MICROARCH.md): the fabric is point to point


def rehearse_exchange(device, ms_per_step, table, dense_bytes, world=8, reps=10):
    """What a rank does AFTER the last backward kernel of a data-parallel step at N = `world`, timed on this one GPU with the sizes
    of N ranks: the deterministic apply of ALL ranks' lookup rows -- sort of world x B x L (id, position)
    pairs + segmented sum of world x 10.5 MB of gradient rows into the table gradient (distributed.SparseRowExchange.exchange
    -> ops.scatter_rows_sorted) -- beside the same apply at one rank's size (what the N = 1 step already contains).  The wire
    time cannot be measured here; it is ESTIMATED from the link rate for a direct exchange over the 7 point-to-point xGMI
    links (every peer's share travels on its own link) and reported as an estimate.  -> dict for `comm.rehearsal_8gpu`."""
    from transformers4rec_amd import ops

    V, D = table.shape
    g = torch.Generator(device=device).manual_seed(123)
    d_table = torch.zeros_like(table)

    def timed_apply(n_rows):
        ids = torch.randint(1, V, (n_rows,), device=device, generator=g)
        rows = torch.randn(n_rows, D, device=device, generator=g) * 1e-3
        for _ in range(2):
            ops.scatter_rows_sorted(d_table, ids, rows, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.scatter_rows_sorted(d_table, ids, rows, 0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t1 = timed_apply(BATCH * SEQ)
    tN = timed_apply(world * BATCH * SEQ)
    per_rank_rows_bytes = BATCH * SEQ * (8 + 4 * D)
    link = XGMI_LINK_GBS * 1e9
    # direct exchange: each peer's block arrives on its own link, all (world - 1) links busy at once
    allgather_ms = 1e3 * per_rank_rows_bytes / link
    tables_bytes = V * D * 4
    # reduce-scatter + all-gather of the table bucket, 1 / world of it per peer and link, both phases
    tables_allreduce_ms = 1e3 * 2.0 * (tables_bytes / world) / link
    dense_allreduce_ms = 1e3 * 2.0 * (dense_bytes / world) / link + 0.03          # + a latency floor of ~30 us (two small phases)
    exposed = (tN - t1) + allgather_ms + dense_allreduce_ms
    # the table all-reduce runs under the body's backward; a resident collective costs the kernels next to it 1.25x for its
    # duration even with the CU budget (tools/occupier_curve.py, profiles/r05_h_occupier_budget.json)
    overlap_penalty = 0.25 * tables_allreduce_ms
    proj = ms_per_step + exposed + overlap_penalty
    return {"what": f"one GPU, sizes of {world} ranks: MEASURED apply of all ranks' lookup rows; ESTIMATED wire times (direct exchange over "
                    f"{world - 1} xGMI links at {XGMI_LINK_GBS:.0f} GB/s each); the projection is arithmetic on those, not a measurement",
            "apply_rows_1_rank_ms": round(t1, 4), f"apply_rows_{world}_ranks_ms": round(tN, 4),
            "rows_gathered_bytes": world * per_rank_rows_bytes,
            "estimated_allgather_ms": round(allgather_ms, 4), "estimated_tables_allreduce_ms_overlapped": round(tables_allreduce_ms, 4),
            "estimated_dense_allreduce_ms": round(dense_allreduce_ms, 4),
            "exposed_after_backward_ms": round(exposed, 4), "overlap_penalty_ms": round(overlap_penalty, 4),
            "projected_ms_per_step": round(proj, 4), "projected_sessions_per_s": round(BATCH * world / (proj * 1e-3), 1),
            "projected_scaling_vs_1gpu": round(world * ms_per_step / proj, 2),
            "note": "to be checked against the first real SCALE run; the north_star asks >= 6x at 8 GPUs"}


# --------------------------------------------------------------------------------------------- live HBM traffic
def live_head_traffic(n_rows, timeout_s=150):
    """HBM bytes per launch of the head kernels from the PMC counters, measured IN THIS RUN when rocprofv3 is present: two
    separate passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over tools/pmc_head_workload.py, `--kernel-trace` only
    beside `--pmc`; each counter is calibrated on a copy of known size in the same access pattern inside the same pass (on
    gfx950 FETCH_SIZE reports half the bytes of a wide streaming read: MI355X_MICROARCH.md "HBM" -- the calibration absorbs
    it).  Returns None when rocprofv3 is missing or a pass fails; the caller then quotes the committed figure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="t4r_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "r", "--",
                                sys.executable, os.path.join(ROOT, "tools", "pmc_head_workload.py"), str(n_rows)],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            per = {}
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] == ctr:
                        per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            pick = lambda key: next((v for k, v in per.items() if key in k), None)
            cal = pick("dropout_kernel")
            if not cal:
                return None
            unit = float(1 << 30) / (sum(cal[-3:]) / len(cal[-3:]))
            for key in ("head_fwd_dx_kernel", "head_dw_split_kernel", "split_w_images_kernel", "head_fdx_finalize_kernel"):
                v = pick(key)
                if v:
                    out.setdefault(key, {})[ctr] = int(unit * sum(v[-3:]) / len(v[-3:]))
            out.setdefault("calibration", {})[ctr + "_unit_bytes"] = round(unit, 2)
        except Exception:  # noqa: BLE001
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "head_fwd_dx_kernel" not in out or len(out["head_fwd_dx_kernel"]) != 2:
        return None
    for k, v in out.items():
        if k != "calibration":
            v["bytes"] = v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)
    return out


# --------------------------------------------------------------------------------------------- launch
def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv=None):
    """`python bench.py --gpus N` without a launcher: re-execute this file as N ranks of ONE node under
    torch.distributed.run (rendezvous on 127.0.0.1: the container hostname may not resolve).  stdout / stderr are
    inherited, so rank 0's JSON line is this process's output; returns the launcher's exit code."""
    import subprocess

    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this platform (RCCL)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)]
    cmd += list(sys.argv[1:] if argv is None else argv)
    return subprocess.call(cmd, env=env)


class _StubModel(torch.nn.Module):
    """T4R_BENCH_STUB=1 only: a CPU stand-in with the HIP model's gradient plumbing (a tied table whose lookup
    gradient goes through the row-sparse sink, a dense block carrying the head-backward hook), so that the LAUNCH
    and data-parallel wiring of this file -- self_launch, setup_data_parallel, make_train_step, timed_region, the
    JSON contract -- can be run end to end on a box without GPUs (tests/test_bench_launch_cpu.py).  Its output is
    marked `"stub": true` and is not a measurement."""

    def __init__(self, V=64, D=8):
        super().__init__()
        self.table = torch.nn.Parameter(torch.randn(V, D) * 0.1)
        self.transformer_block = torch.nn.Linear(D, D)

    def forward(self, x, training=True):
        ids = x["item_id"]
        emb = self.table.detach()[ids].requires_grad_()            # lookups are not tracked by autograd, as on HIP
        h = self.transformer_block(emb)
        labels = ids.reshape(-1)
        loss = torch.nn.functional.cross_entropy(h.reshape(-1, h.shape[-1]) @ self.table.t(), labels)

        def scatter(g):
            sink = getattr(self.table, "_t4r_sparse_sink", None)
            rows = g.reshape(-1, g.shape[-1])
            if sink is not None:
                sink.add_rows(self.table, labels, rows, padding_idx=0)
            else:
                self.table.grad.index_add_(0, labels, rows)
        emb.register_hook(scatter)
        return {"loss": loss, "labels": labels}


def _stub_main(args, world, rank):
    """the N-rank flow of main() on CPU tensors over gloo (see _StubModel)"""
    import torch.distributed as dist

    import transformers4rec_amd as tr
    from transformers4rec_amd.optim import FlatParams

    if world > 1:
        dist.init_process_group("gloo")
    torch.manual_seed(0)
    model = _StubModel()
    tables = FlatParams([("table", model.table)])
    dense = FlatParams([(n, p) for n, p in model.transformer_block.named_parameters()])

    def cpu_apply(d_table, ids, rows, padding_idx):
        keep = ids != padding_idx
        d_table.index_add_(0, ids[keep], rows[keep])

    def setup(mode):
        if mode == "dense":
            return tr.GradReducer(dense.grad, tables.grad, sparse=None), None
        sparse = tr.SparseRowExchange(apply_fn=cpu_apply, equal_sizes=True).attach(model.table)
        reducer = tr.GradReducer(dense.grad, tables.grad, sparse=sparse)
        return reducer, (tr.head_backward_hook(model, reducer.reduce_tables_async) if world > 1 else None)

    def teardown(hook):
        if hook is not None:
            hook.remove()
        tr.SparseRowExchange.detach(model.table)

    class _Sgd:
        def step(self, grad_scale=1.0):
            for f in (dense, tables):
                f.data.add_(f.grad, alpha=-0.05 * grad_scale)
                f.grad.zero_()

    g = torch.Generator().manual_seed(1000 + rank)
    batches = [{"item_id": torch.randint(1, 64, (16, 5), generator=g)} for _ in range(4)]
    exchange_mode, exchange_ms = table_exchange_mode() if world > 1 else "local", None
    if exchange_mode == "auto":
        exchange_mode, reducer, _hook, exchange_ms = pick_table_exchange(setup, teardown, lambda red: make_train_step(model, batches, red, _Sgd()),
                                                                         world, "cpu", steps=2, warm=1)
    else:
        reducer, _hook = setup("sparse" if exchange_mode == "local" else exchange_mode)
    if world > 1 and dist.get_world_size() != args.gpus:
        raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    step = make_train_step(model, batches, reducer, _Sgd())
    dt, out, n_lab = timed_region(step, args.warmup, args.steps, world, "cpu")
    windows = extra_windows(step, args.steps, world, "cpu", args.warmup + args.steps, n_windows=2)
    if rank == 0:
        print(json.dumps({"metric": "launch-plumbing stub (no GPU, not a measurement)", "stub": True,
                          "value": round(16 * world * args.steps / dt, 1), "unit": "sessions/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "ms_per_step_windows": windows, "comm": {"table_exchange_ms_per_step": exchange_ms},
                          "data": "synthetic", "config": {"workload": "stub", "global_batch": 16 * world,
                                                          "parallelism": f"dp{world}", "table_exchange": exchange_mode,
                                                          "world_size": world,
                                                          "final_loss": round(float(out["loss"].detach()), 4),
                                                          "label_rows": n_lab}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preheat-seconds", type=float, default=5.0,
                    help="untimed steps run before the W warm-up steps until the device clocks have ramped "
                         "(a fresh process measured 7.7 ms/step in its first second, 7.1 ms afterwards)")
    ap.add_argument("--dropout", type=float, default=0.3)  # XLNetConfig.build default (config/transformer.py:442)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="quote the committed PMC figure for roofline.traffic instead of measuring it (two rocprofv3 --pmc passes)")
    ap.add_argument("--extra-streams", type=int, default=0,
                    help="measurement only: K more HIP streams, each with one tiny launch per step -- how the step reacts to "
                         "more streams than hardware queues (the data-parallel run adds the collective's stream)")
    ap.add_argument("--config", choices=("c2", "c3"), default="c2",
                    help="c2 = BASELINE.json configs[1] (the configuration `metric` is quoted on; default); "
                         "c3 = configs[2], the multi-feature input block on the same XLNet (the 8-GPU DP configuration)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called without a launcher: become the launcher (one rank per GPU of this node)
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if os.environ.get("T4R_BENCH_STUB", "0") == "1":
        return _stub_main(args, world, rank)
    # T4R_BENCH_BACKEND=gloo (with T4R_BENCH_SHARE_GPU=1: all ranks on one device) exercises the N > 1 code path of this
    # file on a single-GPU box; the measured configuration is always nccl (= RCCL), one rank per GPU
    backend = os.environ.get("T4R_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("T4R_BENCH_SHARE_GPU", "0") == "1" else local_rank
    if world > 1:       # (before the first HIP call: the runtime reads its environment once)
        # The table all-reduce runs UNDER the body's backward, whose token-tile kernels launch one workgroup per CU: every CU an
        # RCCL channel holds costs them a second round (tools/occupier_curve.py, profiles/r05_occupier_curve.json): cap the
        # channels, so that the collective's footprint is known and small -- 16 channels move the 51 MB bucket over 7 xGMI
        # links in ~0.3 ms, well inside the ~1 ms backward.  Overridable from the environment.
        # Streams: the step drives the caller's stream + two weight-gradient streams, the collective's stream is the FOURTH;
        # a fifth active stream costs +1.0 ms per step on this runtime and GPU_MAX_HW_QUEUES=8 does not lift that
        # (profiles/r05_q_stream_count.txt) -- so nothing here raises the queue count, and the forward's id-sort stream is off.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

        # first-contact safety: the collectives really span the N ranks the driver asked for
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    tr, schema, model, dense, tables, opt = build(device, args.dropout, config=args.config)
    from transformers4rec_amd import ops

    exchange_mode = table_exchange_mode() if world > 1 else "local"
    exchange_ms = None
    reducer, _hook = setup_data_parallel(tr, model, dense, tables, world)
    masking = model.input_features.masking
    masking.seed, model.transformer_block.transformer.seed = rank_seeds(rank)
    # synthetic Schema-driven sessions, resident in HBM before the timed region (8 distinct batches)
    batches = [tr.random_data_from_schema(schema, BATCH, SEQ, seed=1000 * rank + i, device=device)
               for i in range(8)]
    model.train()
    train_step = make_train_step(model, batches, reducer, opt)

    # device pre-heat: full training steps for a fixed TIME (clock ramp), then the model / optimizer
    # state is rolled back, so the measured run (and its final loss) does not depend on how many
    # pre-heat steps this particular box managed
    snap = ([f.data.clone() for f in opt.flats], [(m.clone(), v.clone()) for m, v in opt.state], opt.step_count,
            tr.get_rng_state(model))
    if world > 1 and exchange_mode == "auto":
        # both forms of the table-gradient exchange timed on this fabric (inside the rolled-back pre-heat), the faster kept
        teardown_data_parallel(tables, _hook)
        exchange_mode, reducer, _hook, exchange_ms = pick_table_exchange(
            lambda m: setup_data_parallel(tr, model, dense, tables, world, mode=m),
            lambda hk: teardown_data_parallel(tables, hk),
            lambda red: make_train_step(model, batches, red, opt), world, device)
        train_step = make_train_step(model, batches, reducer, opt)
    if args.extra_streams > 0:
        _extra = [torch.cuda.Stream(device) for _ in range(args.extra_streams)]
        _tick = [torch.zeros(256, device=device) for _ in _extra]
        _plain_step = train_step

        def train_step(i):
            for st, t in zip(_extra, _tick):
                with torch.cuda.stream(st):
                    t.add_(1.0)
            return _plain_step(i)
    t_pre = time.perf_counter()
    n_pre = 0
    go = torch.ones(1, device=device, dtype=torch.int32)
    while True:
        # N > 1: every rank must run the SAME number of pre-heat steps (each step holds collectives), so the ranks agree
        # on continuing: stop as soon as any rank's clock says so
        go.fill_(1 if time.perf_counter() - t_pre < args.preheat_seconds else 0)
        if world > 1:
            dist.all_reduce(go, op=dist.ReduceOp.MIN)
        if int(go.item()) == 0:
            break
        train_step(0)
        torch.cuda.synchronize()
        n_pre += 1
    preheat_s = time.perf_counter() - t_pre
    for f, d in zip(opt.flats, snap[0]):
        f.data.copy_(d)
    for (m, v), (m0, v0) in zip(opt.state, snap[1]):
        m.copy_(m0)
        v.copy_(v0)
    opt.step_count = snap[2]
    tr.set_rng_state(model, snap[3])
    del snap
    dt, out, n_lab = timed_region(train_step, args.warmup, args.steps, world, device)
    loss = float(out["loss"].detach())
    windows = extra_windows(train_step, args.steps, world, device, args.warmup + args.steps)

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream
    # (the same GEMM launches the timed steps issue: logits = X[N_m,128] @ W[100001,128]^T)
    N_m = max(1, n_lab // max(1, args.steps))
    W = model.input_features.item_embedding_table.weight.detach()
    xr = torch.randn(N_m, D_MODEL, device=device)
    ld = ops.pad_ld(W.shape[0])
    buf = torch.empty((N_m, ld), device=device)

    def timed(fn, reps=20):
        """average device time of `fn` over `reps` back-to-back launches (HIP events on the launch stream)"""
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # the step runs in the library's default precision mode ("auto": fp32-class accuracy -- the head and the body's token-tile
    # kernels on the two-way fp16 split with power-of-two scales, three matrix-core products per fp32-equivalent one; the
    # general GEMM's remaining large contractions on the exact 3-way bf16 split, six products); timed in both forms
    mode = ops.get_precision()
    from transformers4rec_amd.prediction_task import _head_split_ok

    general_ms = timed(lambda: ops.gemm(xr, W, False, True, out=buf[:, : W.shape[0]]))
    head_split = _head_split_ok(xr, W, N_m, W.shape[0])
    fdx = False
    logits_only_ms = None
    if head_split:      # the launches the step issues at this shape: csrc/head_split.hip
        hws = ops.head_split_prepare(xr, W.shape[0])
        logits_only_ms = timed(lambda: ops.call("t4r_head_split_logits_ce", ops._stream(), hws.data_ptr(), W.data_ptr(), W.stride(0),
                                                buf.data_ptr(), ld, None, None, None, None, N_m, W.shape[0], D_MODEL, 1.0, 0.0, None))
        gemm_ms = logits_only_ms
        fdx = bool(ops.head_split_fdx_supported(D_MODEL))
        if fdx:
            # round 5: the step's dominant kernel is the ONE-PASS forward (logits + CE statistics + d X: head_fwd_dx_kernel).
            # One full call fills the workspace (table maximum, table images), then the kernel alone is timed (labels = NULL)
            lab_ = torch.randint(1, W.shape[0], (N_m,), device=device)
            ops.head_split_logits_ce_dx(hws, xr, W, lab_, ldc=ld)
            gemm_ms = timed(lambda: ops.call("t4r_head_split_logits_ce_dx", ops._stream(), hws.data_ptr(), xr.data_ptr(), xr.stride(0),
                                             W.data_ptr(), W.stride(0), buf.data_ptr(), ld, None, None, None, None, None, 0, None,
                                             N_m, W.shape[0], D_MODEL, 1.0, 0.0, None))
    else:
        gemm_ms = general_ms
    with ops.precision("fp32"):
        gemm_ms_f32 = timed(lambda: ops.gemm(xr, W, False, True, out=buf[:, : W.shape[0]]))
    n_contractions = 2.0 if fdx else 1.0       # the one-pass forward runs the logits AND the d X contraction
    flops = n_contractions * 2.0 * N_m * W.shape[0] * D_MODEL
    split = mode in ("auto", "fp32_bf16x3")
    # matrix instructions per fp32-equivalent one: 6 (three bf16 planes) or, in csrc/head_split.hip's forward, 3 (two-way fp16 split)
    n_prod = float(_lib_int("t4r_head_split_fwd_products")) if (split and head_split) else (6.0 if split else 1.0)
    executed = n_prod * flops
    peak = {"fp32": MFMA_F32_PEAK_TFLOPS, "auto": MFMA_BF16_PEAK_TFLOPS, "fp32_bf16x3": MFMA_BF16_PEAK_TFLOPS,
            "bf16": MFMA_BF16_PEAK_TFLOPS, "fp16": MFMA_BF16_PEAK_TFLOPS}[mode]
    achieved = executed / (gemm_ms * 1e-3) / 1e12
    # algorithmic bytes: X and W read once, the [N, V] logits written once (+ the d X rows written once by the one-pass form)
    alg_bytes = 4.0 * (N_m * D_MODEL + W.shape[0] * D_MODEL + N_m * W.shape[0]) + (4.0 * N_m * D_MODEL if fdx else 0.0)
    alg_gbs = alg_bytes / (gemm_ms * 1e-3) / 1e9
    # which roof: the one the ALGORITHMIC work takes longer to cross (bytes and flops are algorithmic minima)
    alg_tfs = flops / (gemm_ms * 1e-3) / 1e12
    hbm_bound = alg_bytes / (HBM_PEAK_GBS * 1e9) > flops / (peak * 1e12)
    # embedding gather (HBM bound): bytes = T * (8 id + 512 row read + 512 row write)
    ids = batches[0]["item_id"]
    feats = [dict(kind=0, input=ids, table=W, dim=D_MODEL, col=0, rows=W.shape[0])]

    def graph_timed(fn, reps=50):
        """as `timed`, with the launches replayed from one HIP graph: a 5 us kernel is otherwise
        measured at the host's launch rate (ctypes call ~10 us), not at its own duration.
        N > 1: no capture next to a live RCCL communicator (its watchdog thread may touch the device
        during the capture) -- the plain event loop is used there, and on any capture failure."""
        if world > 1:
            return timed(fn, reps)
        try:
            fn()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                fn()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    for _ in range(reps):
                        fn()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        except Exception as exc:      # noqa: BLE001 - measurement aid only; never fail the bench line on it
            print(f"[bench] graph replay unavailable ({type(exc).__name__}: {exc}); timing with the event loop",
                  file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            return timed(fn, reps)

    gather_ms = graph_timed(lambda: ops.seq_features_fwd(feats, "concat", BATCH, SEQ, SEQ, D_MODEL))
    gather_bytes = BATCH * SEQ * (8 + 4 * D_MODEL + 4 * D_MODEL)
    gather_gbs = gather_bytes / (gather_ms * 1e-3) / 1e9
    # the same kernel on the tokens of the GLOBAL batch (8192 sessions, the size north_star quotes the
    # gather target on), against a table that does NOT fit the caches (10 M rows x 128 = 5.1 GB, fresh ids
    # per launch position): at 20 480 tokens (21 MB) the launch ramp is a third of the kernel
    GB = 8192
    big_rows = 10_000_001
    Wbig = torch.empty((big_rows, D_MODEL), device=device).normal_()
    ids_g = [torch.randint(1, big_rows, (GB, SEQ), device=device) for _ in range(4)]
    feats_g = [[dict(kind=0, input=t, table=Wbig, dim=D_MODEL, col=0, rows=big_rows)] for t in ids_g]
    state = {"k": 0}

    def gather_big():
        state["k"] += 1
        ops.seq_features_fwd(feats_g[state["k"] % 4], "concat", GB, SEQ, SEQ, D_MODEL)

    gather_g_ms = graph_timed(gather_big, reps=20)
    gather_g_bytes = GB * SEQ * (8 + 4 * D_MODEL + 4 * D_MODEL)
    gather_g_gbs = gather_g_bytes / (gather_g_ms * 1e-3) / 1e9
    # configs[2] (C3): the 336-wide concatenation -- item 128 + three categoricals 64 + two soft embeddings 8 (dense rows) --
    # at the global batch, item table out of cache, id sets rotated
    c3_state = {"k": 0}
    cats = [torch.empty((card, 64), device=device).normal_() for card in (1001, 501, 101)]
    # (not `dense`: that name is the flat parameter bucket the comm report below reads at N > 1 -- the shadowing crashed every
    #  N > 1 run after the probes; found by the two-rank run of tests/test_distributed_gpu.py::test_bench_main_two_ranks)
    dense_rows = [torch.randn(GB * SEQ, 8, device=device) for _ in range(2)]
    feats_c3 = []
    for t in ids_g:
        f = [dict(kind=0, input=t, table=Wbig, dim=D_MODEL, col=0, rows=big_rows)]
        col = D_MODEL
        for tab in cats:
            f.append(dict(kind=0, input=t % tab.shape[0], table=tab, dim=64, col=col, rows=tab.shape[0]))
            col += 64
        for dn in dense_rows:
            f.append(dict(kind=1, input=dn, table=None, dim=8, col=col, rows=0))
            col += 8
        feats_c3.append(f)
    W3 = D_MODEL + 3 * 64 + 2 * 8

    def gather_c3():
        c3_state["k"] += 1
        ops.seq_features_fwd(feats_c3[c3_state["k"] % 4], "concat", GB, SEQ, SEQ, W3)

    gather_c3_ms = graph_timed(gather_c3, reps=20)
    gather_c3_bytes = GB * SEQ * (4 * 8 + 4 * W3 + 4 * W3)
    gather_c3_gbs = gather_c3_bytes / (gather_c3_ms * 1e-3) / 1e9
    # the byte-moving ceiling of THIS box, in the same run: a device-to-device copy of 1 GiB (read + write = 2 GiB of HBM
    # traffic) by (a) torch's copy_ and (b) a plain float4 copy kernel (tools/t4r_tools.hip; the form MI355X_MICROARCH.md
    # quotes at 6.29 TB/s), plain and non-temporal -- the best of them is the rate a kernel that only moves bytes reaches here
    cp_src = Wbig.view(-1)[: (1 << 28)]
    cp_dst = torch.empty_like(cp_src)
    copy_ms = graph_timed(lambda: cp_dst.copy_(cp_src), reps=10)
    copy_rates = {"torch_copy_": 2.0 * cp_src.numel() * 4 / (copy_ms * 1e-3) / 1e9}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import t4r_tools

        for name, cmode in (("float4_kernel", 0), ("float4_kernel_nt", 1)):
            ms_ = graph_timed(lambda m=cmode: t4r_tools.copy(cp_dst, cp_src, mode=m), reps=10)
            copy_rates[name] = 2.0 * cp_src.numel() * 4 / (ms_ * 1e-3) / 1e9
    except Exception as exc:      # noqa: BLE001 - measurement aid only
        copy_rates["float4_kernel_error"] = f"{type(exc).__name__}: {exc}"
    copy_gbs = max(v for v in copy_rates.values() if isinstance(v, float))
    del Wbig, ids_g, feats_g, feats_c3, cats, dense_rows, cp_src, cp_dst

    # ---- the transformer body's fused kernels (csrc/xlnet_fused*.hip), timed live at this run's shape: the feed-forward
    # block forward / backward (one launch each; 2 * T * 4D * D * 2 algorithmic flops per direction; `npb` matrix-core
    # products executed per algorithmic product: 3 in the shipped two-way fp16 split, 6 on the bf16 planes)
    body = None
    if ops.xlnet_fused_supported(D_MODEL):
        Tt = BATCH * SEQ
        lay = model.transformer_block.transformer.layer[0]
        prm = [q.detach().contiguous() for q in lay.ordered_params()]
        planes = ops.xlnet_layer_prepare(prm, D_MODEL)
        h1 = torch.randn(Tt, D_MODEL, device=device)
        dyy = torch.randn(Tt, D_MODEL, device=device)
        _, sv = ops.xlnet_ff_fwd(h1, planes, prm[10], prm[12], prm[13], prm[14], 0.03, args.dropout, 7, 11, 12)
        zz = lambda k: torch.zeros(k, device=device)
        gbuf = (zz(D_MODEL), zz(D_MODEL), zz(D_MODEL), zz(4 * D_MODEL))
        ff_fwd_ms = timed(lambda: ops.xlnet_ff_fwd(h1, planes, prm[10], prm[12], prm[13], prm[14], 0.03, args.dropout, 7, 11, 12))
        ff_bwd_ms = timed(lambda: ops.xlnet_ff_bwd(dyy, h1, sv, prm[13], planes, *gbuf, args.dropout, 7, 11, 12))
        ff_flops = 2.0 * Tt * 4 * D_MODEL * D_MODEL * 2
        npb = float(_lib_int("t4r_xlnet_fused_products"))       # 3: two-way fp16 split, 6: three bf16 planes
        body = {"kernel": "xlnet_ff_fwd_kernel<128, 5> / xlnet_ff_bwd_kernel<128, 5> (token-tile-stationary feed-forward block, "
                          "one launch per direction; " + ("fp32-class two-way fp16 split: per-token / per-matrix power-of-two scales, "
                          "three products" if npb == 3.0 else "fp32-accurate three-plane bf16 products, six products") + ")",
                "bound": "mfma + valu (measured: VALU work does not issue under matrix instructions on this chip, "
                         "tools/mfma_valu_overlap.hip)",
                "fwd_avg_launch_ms": round(ff_fwd_ms, 4), "bwd_avg_launch_ms": round(ff_bwd_ms, 4),
                "flops_per_launch": ff_flops, "executed_flops_per_launch": npb * ff_flops,
                "fwd": {"achieved": round(npb * ff_flops / (ff_fwd_ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(npb * ff_flops / (ff_fwd_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                        "fp32_equivalent_TFLOPs": round(ff_flops / (ff_fwd_ms * 1e-3) / 1e12, 1)},
                "bwd": {"achieved": round(npb * ff_flops / (ff_bwd_ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(npb * ff_flops / (ff_bwd_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                        "fp32_equivalent_TFLOPs": round(ff_flops / (ff_bwd_ms * 1e-3) / 1e12, 1),
                        "note": "includes the two partial-sum reductions of d gamma, d beta, d b2, d b1"},
                "algorithmic_bytes_fwd": int(4 * Tt * D_MODEL * (2 + 8 + 1 + 1)), "algorithmic_bytes_bwd": int(4 * Tt * D_MODEL * (3 + 4 + 1 + 4 + 2)),
                "traffic_fwd": 27.0e6 + 105.1e6, "traffic_bwd": 88.5e6 + 72.6e6,
                "traffic_source": "committed (not measured in this run): profiles/r03_e_pmc_gemm_fetch_write.csv "
                                  "(FETCH_SIZE x 2 KB + WRITE_SIZE KB, separate rocprofv3 --pmc passes)",
                "matrix_pipe_busy": {"fwd": 0.125, "bwd": 0.089, "source": "profiles/r04_f_pmc_mfma_busy.csv (in-step, the shipped two-way "
                                     "fp16 form: SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs; the six-product bf16 "
                                     "form measured 0.218 / 0.169 in round 3 -- half the matrix instructions, same wall time)"}}
        # the attention half of the forward as ONE kernel (csrc/xlnet_attn_block.hip, round 4), timed live at this shape;
        # north_star's "MFMA utilisation on attention" is the committed in-step counter figure beside it
        if ops.xlnet_attn_block_supported(SEQ, D_MODEL, N_HEAD):
            gen = torch.Generator(device=device).manual_seed(5)
            kr_ = 0.3 * torch.randn(BATCH * 2 * SEQ, D_MODEL, device=device, generator=gen)
            hh_ = torch.randn(Tt, D_MODEL, device=device, generator=gen)
            dh_ = D_MODEL // N_HEAD
            ab = lambda: ops.xlnet_attn_block_fwd(hh_, planes, prm[3].view(D_MODEL, D_MODEL), kr_, prm[5].view(-1), prm[6].view(-1),
                                                  prm[7], prm[8], BATCH, SEQ, N_HEAD, 0.03, args.dropout, 7, 21, 22)
            ab_ms = timed(ab)
            ab_flops = 2.0 * Tt * D_MODEL * 4 * D_MODEL + 2.0 * BATCH * N_HEAD * (2 * SEQ * SEQ * dh_ + SEQ * 2 * SEQ * dh_)
            body["attention_block_fwd"] = {
                "kernel": "xlnet_attn_block_fwd_kernel<128, 32> (q|k|v projection + relative attention core + o-projection + dropout + "
                          "residual + LayerNorm in one launch; exact fp32 products on v_mfma_f32_16x16x4_f32)",
                "avg_launch_ms": round(ab_ms, 4), "flops_per_launch": ab_flops,
                "achieved": round(ab_flops / (ab_ms * 1e-3) / 1e12, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ab_flops / (ab_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                # h in; q | k | v, attn_vec, o-projection, h1 out (7 T D floats) + the per-session k_r rows read (B 2L D floats)
                "algorithmic_bytes": int(4 * Tt * D_MODEL * 7 + 4 * BATCH * 2 * SEQ * D_MODEL),
                "traffic": int(12616 * 2 * 1024 + 61920 * 1024), "traffic_unit": "bytes/launch",
                "traffic_source": "committed (not measured in this run): profiles/r04_i_pmc_attn_block_fetch_write.txt (rocprofv3 --pmc "
                                  "FETCH_SIZE x 2 KB, WRITE_SIZE KB, separate passes, tools/pmc_kernel.sh over tools/attn_block_bench.py --once)",
                "matrix_pipe_busy": {"fwd": 0.378, "bwd_core": 0.311, "target": 0.50,
                                     "source": "profiles/r04_f_pmc_mfma_busy.csv (in-step; bwd_core = xlnet_attn_mfma_bwd_kernel, unchanged "
                                               "since round 1; round 3: forward core 0.21)"}}
            del kr_, hh_
        del planes, h1, dyy, sv

    # HBM bytes of the launches from the PMC counters (FETCH_SIZE x2 on gfx950, WRITE_SIZE calibrated
    # on a known copy; collected in separate rocprofv3 --pmc passes and committed under profiles/).
    # It is a property of the kernel + shape, not of this run: taken from the committed measurement.
    def committed(name):
        for rnd in ("r05", "r02", "r01_g"):
            path = os.path.join(ROOT, "profiles", f"{rnd}_{name}.json")
            if os.path.exists(path):
                with open(path) as f:
                    return json.load(f)
        return None

    traffic, traffic_src, traffic_detail = None, None, None
    if fdx and world == 1 and not args.no_live_traffic:
        torch.cuda.synchronize()
        lt = live_head_traffic(N_m)
        if lt is not None:
            traffic = lt["head_fwd_dx_kernel"]["bytes"]
            traffic_detail = lt
            traffic_src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (--kernel-trace only "
                           "beside them) over tools/pmc_head_workload.py at this run's label-row count; each counter calibrated on a "
                           "1 GiB streaming copy inside the same pass (FETCH_SIZE counts half the bytes of wide reads on gfx950)")
    tj = committed("pmc_traffic")
    if tj is not None and ("head_fwd_dx" in tj.get("kernel", "")) != fdx:
        tj = None                   # the committed counters describe the other form of the head's forward
    if traffic is None and tj is not None:
        traffic = int(tj["traffic_bytes_per_launch"] * N_m / tj["n_rows"])   # scales with the label rows
        traffic_src = "committed (not measured in this run; scaled by the label rows): " + tj["source"]
    gj = committed("pmc_traffic_gather")

    # bytes every rank hands to the collectives per step (payload, not wire traffic: a ring all-reduce moves
    # 2 (N-1)/N of it per link): the dense bucket, the tables bucket (tied head: dense d W) and the row-sparse exchange
    if fdx:
        kernel_desc = ("head_fwd_dx_kernel<4> (round 5, ONE pass over (128-row tile x item range) workgroups: next-item logits X@W^T "
                       "stored once + the softmax statistics of the loss + d X = sum_v p_v W_v from the same score tiles in registers, "
                       "flash-attention style running reference; fp32-class accuracy: two-way fp16 split with power-of-two scales, three "
                       "v_mfma_f32_32x32x16_f16 products per K=16 in BOTH contractions; the backward then reads the logits once (d W) "
                       "instead of twice)")
    elif head_split and n_prod == 3.0:
        kernel_desc = ("head_logits_ce_kernel<4, fp16x2> (next-item logits X@W^T + the softmax statistics of the loss; fp32-class "
                       "accuracy: two-way fp16 split with power-of-two tensor scales, three v_mfma_f32_32x32x16_f16 products per K=16; "
                       "W fragments register-resident, X plane blocks through LDS)")
    elif head_split:
        kernel_desc = ("head_logits_ce_kernel<4> (next-item logits X@W^T + the softmax statistics of the loss; fp32-accurate: exact "
                       "3-way bf16 split, six v_mfma_f32_32x32x16_bf16 products per K=16; W fragments register-resident, X plane "
                       "blocks through LDS)")
    elif split:
        kernel_desc = ("gemm_f32_kernel<128,64,32,NT,PREC=1> (next-item logits X@W^T; fp32-accurate: exact 3-way bf16 split, six "
                       "v_mfma_f32_32x32x16_bf16 products per K=16)")
    else:
        kernel_desc = "gemm_f32_kernel<128,128,16,NT> (next-item logits X@W^T)"
    sparse = getattr(reducer, "sparse", None)
    comm = {"dense_bucket_bytes": int(dense.grad.numel() * 4) if world > 1 else 0,
            "tables_bucket_bytes": int(tables.grad.numel() * 4) if world > 1 and reducer.tables is not None else 0,
            "row_sparse_gathered_bytes": int((sparse.bytes_exchanged if sparse is not None else 0) //
                                             max(1, n_pre + args.warmup + args.steps * (1 + windows["n_windows"]))),
            "table_exchange_ms_per_step": exchange_ms,
            "note": "per rank and step; the tables all-reduce starts right after the head's backward and runs under the "
                    "transformer body's backward; the dense bucket and the (ids, rows) all-gather are exposed"}
    if rank == 0:
        res = {
            "metric": "training sessions/sec (XLNet 4x128, 100k items, seq 20, MLM, tied full softmax" +
                      (", multi-feature input: item + 3 categoricals + 2 SoftEmbedding, concat" if args.config == "c3" else "") + ") + Recall@20",
            "value": round(BATCH * world * args.steps / dt, 1), "unit": "sessions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config],
                       "global_batch": BATCH * world, "seq_len": SEQ, "parallelism": f"dp{world}",
                       "dropout": args.dropout, "label_rows_per_step": N_m, "final_loss": round(loss, 4),
                       "head_mode": model.prediction_task.resolve_head_mode(N_m, W.shape[0]),
                       "precision_mode": mode, "table_exchange": exchange_mode, "world_size": world,
                       "collective_backend": (backend if world > 1 else None),
                       "nccl_max_nchannels": (os.environ.get("NCCL_MAX_NCHANNELS") if world > 1 else None),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "extra_streams": args.extra_streams,
                       "prefetch": int(os.environ.get("T4R_PREFETCH", "1")),
                       "preheat_s": round(preheat_s, 2), "preheat_steps": n_pre,
                       "timed_region_s": round(dt, 4)},
            "ms_per_step_windows": windows,
            "roofline": {"kernel": kernel_desc,
                         # `achieved` / `frac` are ALGORITHMIC work over the live launch time against the roof that algorithmic work
                         # crosses last (an executed-flop fraction moves with the arithmetic scheme, not with the
                         # kernel's speed).  At configs[1] the one-pass head moves 1.17 GB for 142 GF: 122 flop/B against a machine
                         # balance of 312 -- writing the logits once is its roof.  The matrix side is reported beside it in both
                         # currencies: algorithmic (fp32-equivalent flops against the fp16 peak) and executed (x3: the two-way fp16
                         # split that gives fp32-class accuracy), and against the scheme's own ceiling peak / 3
                         "bound": "hbm" if hbm_bound else "mfma", "precision_mode": mode,
                         "achieved": round(alg_gbs, 1) if hbm_bound else round(alg_tfs, 2),
                         "peak": HBM_PEAK_GBS if hbm_bound else peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                         "frac": round(alg_gbs / HBM_PEAK_GBS, 4) if hbm_bound else round(alg_tfs / peak, 4),
                         "arithmetic_intensity_flop_per_byte": round(flops / alg_bytes, 1),
                         "executed_intensity_flop_per_byte": round(executed / alg_bytes, 1),
                         "machine_balance_flop_per_byte": round(peak * 1e12 / (HBM_PEAK_GBS * 1e9), 1),
                         "note": "achieved = algorithmic " + ("bytes (X + W read once, the [N, V] logits and the d X rows written once)"
                                 if hbm_bound else "flops") + " per launch / average launch time, measured in this run with HIP events on "
                                 "the launch stream; the roof is the one the algorithmic work needs longer to cross",
                         "mfma_side": {"achieved": round(alg_tfs, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(alg_tfs / peak, 4),
                                       "executed_achieved": round(achieved, 2), "executed_frac": round(achieved / peak, 4),
                                       "products_per_fp32_equivalent": int(n_prod),
                                       "frac_of_scheme_ceiling": round(alg_tfs / (peak / n_prod), 4),
                                       "note": f"algorithmic = {int(n_contractions)} x 2*N*V*D fp32-equivalent flops; executed = x{int(n_prod)} matrix "
                                               "instructions (two-way fp16 split, fp32 accumulation: 3-5e-6 of the largest output against fp64 "
                                               "in the tests); both against the dense fp16 MFMA peak, the scheme's ceiling is peak / 3"},
                         "contractions_per_launch": int(n_contractions),
                         "logits_only_kernel_ms": None if logits_only_ms is None else round(logits_only_ms, 4),
                         "fp32_equivalent": {"achieved": round(flops / (gemm_ms * 1e-3) / 1e12, 2),
                                             "peak": MFMA_F32_PEAK_TFLOPS,
                                             "frac": round(flops / (gemm_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
                         "hbm_side": {"algorithmic_GBps": round(alg_gbs, 1), "frac_of_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4),
                                      "note": "the launch writes the [N, V] logits once (1.1 GB): its second bound"},
                         "general_gemm_same_shape_ms": round(general_ms, 4),
                         "fp32_matrix_core_form": {"avg_launch_ms": round(gemm_ms_f32, 4),
                                                   "achieved": round(flops / (gemm_ms_f32 * 1e-3) / 1e12, 2),
                                                   "frac": round(flops / (gemm_ms_f32 * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "traffic_over_algorithmic": None if traffic is None else round(traffic / alg_bytes, 3),
                         "traffic_head_kernels": traffic_detail,
                         "algorithmic_bytes": int(alg_bytes),
                         "avg_launch_ms": round(gemm_ms, 4), "flops_per_launch": flops,
                         "executed_flops_per_launch": executed},
            "roofline_gather": {"kernel": "seq_features_fwd_fast_kernel<32, 2> (embedding gather, streaming stores)", "bound": "hbm",
                                "achieved": round(gather_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(gather_gbs / HBM_PEAK_GBS, 4),
                                "traffic": None if gj is None else gj.get("traffic_bytes_per_launch_c2"),
                                "avg_launch_ms": round(gather_ms, 5), "bytes_per_launch": gather_bytes,
                                "note": "per-GPU batch: 20 480 tokens against the 51 MB table (cache resident)",
                                "measured_copy_GBps": round(copy_gbs, 1),
                                "measured_copy_forms_GBps": {k: (round(v, 1) if isinstance(v, float) else v) for k, v in copy_rates.items()},
                                "measured_copy_note": "1 GiB device-to-device copy in this run (read + write bytes / time), best of torch's "
                                                      "copy_ and the float4 copy kernel of tools/t4r_tools.hip (plain / non-temporal): what "
                                                      "a pure byte-moving kernel reaches on this box; frac_of_measured_copy = gather rate / this",
                                "at_global_batch_8192_out_of_cache": {
                                    "achieved": round(gather_g_gbs, 1), "frac": round(gather_g_gbs / HBM_PEAK_GBS, 4),
                                    "frac_of_measured_copy": round(gather_g_gbs / copy_gbs, 4),
                                    "avg_launch_ms": round(gather_g_ms, 5), "bytes_per_launch": gather_g_bytes,
                                    "table": "10 000 001 x 128 fp32 (5.1 GB), 4 id sets rotated",
                                    "traffic": None if gj is None else gj.get("traffic_bytes_per_launch_8192"),
                                    "traffic_source": None if gj is None else gj.get("source")},
                                "c3_multi_feature_at_global_batch": {
                                    "kernel": "seq_features_fwd_fast_kernel<64, 4, 2> (item 128 + 3 x 64 categorical + 2 x 8 dense rows, "
                                              "concat, 336 floats per token)",
                                    "achieved": round(gather_c3_gbs, 1), "frac": round(gather_c3_gbs / HBM_PEAK_GBS, 4),
                                    "frac_of_measured_copy": round(gather_c3_gbs / copy_gbs, 4),
                                    "avg_launch_ms": round(gather_c3_ms, 5), "bytes_per_launch": gather_c3_bytes,
                                    "note": "163 840 tokens, item table 10 000 001 x 128 (out of cache, 4 id sets rotated), the three "
                                            "small tables are cache resident"}},
        }
        if body is not None:
            res["roofline_body"] = body
        if world == 1 and args.config == "c2":
            try:
                comm["rehearsal_8gpu"] = rehearse_exchange(device, 1e3 * dt / args.steps, W, int(dense.grad.numel() * 4))
            except Exception as exc:      # noqa: BLE001
                comm["rehearsal_8gpu"] = {"error": f"{type(exc).__name__}: {exc}"}
        res["comm"] = comm
    # Recall@20, the other half of the metric.  N > 1: a collective probe, every rank takes part (T4R_BENCH_DP_RECALL=0
    # skips it on all ranks alike)
    probe = None
    if not args.no_recall and world > 1 and os.environ.get("T4R_BENCH_DP_RECALL", "1") == "1":
        del model, opt, dense, tables, reducer, train_step, batches
        probe = recall_probe_dp(device, args.dropout, world, rank, config=args.config)
    if rank == 0:
        if probe is not None:
            res["recall_at_20"] = probe
        if not args.no_recall and world == 1:
            try:
                # (configs[2]'s wider input block leaves the initial loss plateau later, and WHEN depends on the keys: at 600 steps
                #  eight key sets gave Recall@20 0.01 .. 0.86 -- the figure measured the plateau exit, and round 5's 0.842 was one
                #  lucky sample; 1000 steps put most runs behind it.  The spread is in the line either way.)
                res["recall_at_20"] = recall_probe(device, args.dropout, config=args.config,
                                                   train_steps=RECALL_TRAIN_STEPS if args.config == "c2" else 5 * RECALL_TRAIN_STEPS,
                                                   n_seeds=PROBE_SEEDS if args.config == "c2" else 4)
            except Exception as exc:      # noqa: BLE001 - never lose the throughput line to the metric probe
                res["recall_at_20"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.dropout, config=args.config)
            try:        # the reference itself beside the port: run here when its tree exists, else the committed number
                ref = cpu_baseline_reference(args.dropout) if args.config == "c2" else None
            except Exception as exc:      # noqa: BLE001
                ref = {"error": f"{type(exc).__name__}: {exc}"}
            if ref is not None:
                res["cpu_baseline"]["reference_verbatim"] = ref
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
