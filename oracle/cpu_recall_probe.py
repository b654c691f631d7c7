"""TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Recall@20 of the CPU oracle at the BENCHMARKED configuration (VERDICT r3 weak #5,
next #9): BASELINE configs[1] -- 100 001 x 128 table, 4-layer 4-head XLNet, batch 1024, seq 20, MLM p = 0.15, dropout 0.3
(a torch.bernoulli mask per site), tied full softmax, Adam lr 2e-3 -- trained for 200 steps on the Markov-chain sessions of
bench.recall_probe (same chain, same session seeds), evaluated on the same four held-out batches with the reference's
last-item protocol.  The counterpart of bench.py's `recall_at_20.hip_bench_config` (HIP path, same data / steps / lr).  Masks,
dropout draws and the parameter initialisation come from the CPU generator, so the two trajectories are independent samples
of the same training procedure: comparable values, not identical ones.

    python oracle/cpu_recall_probe.py [--steps 200] [--out profiles/r05_cpu_oracle_recall_bench_config.json]
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE, os.path.join(ROOT, "tests")]
import bench  # noqa: E402
import t4r_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_cpu_oracle_recall_bench_config.json"))
    args = ap.parse_args()
    cores = os.cpu_count() or 1
    torch.set_num_threads(min(cores, 32))
    V, D, n, NL, B, L, p = bench.V_ITEMS, bench.D_MODEL, bench.N_HEAD, bench.N_LAYER, bench.BATCH, bench.SEQ, args.dropout
    dh = D // n
    g = torch.Generator().manual_seed(0)
    rn = lambda *s, std=0.01: (std * torch.randn(*s, generator=g)).requires_grad_()
    layers = [dict(q=rn(D, n, dh), k=rn(D, n, dh), v=rn(D, n, dh), o=rn(D, n, dh), r=rn(D, n, dh), r_w_bias=rn(n, dh),
                   r_r_bias=rn(n, dh), ln_w=torch.ones(D, requires_grad=True), ln_b=torch.zeros(D, requires_grad=True),
                   w1=rn(4 * D, D), b1=torch.zeros(4 * D, requires_grad=True), w2=rn(D, 4 * D),
                   b2=torch.zeros(D, requires_grad=True), ff_ln_w=torch.ones(D, requires_grad=True),
                   ff_ln_b=torch.zeros(D, requires_grad=True)) for _ in range(NL)]
    table, memb = rn(V + 1, D, std=0.05), rn(D, std=0.001)
    leaves = [table, memb] + [t for lp in layers for t in lp.values()]
    opt = torch.optim.Adam(leaves, lr=2e-3)
    active = 1 + torch.arange(2000) * (V // 2000)
    keep = lambda *shape: torch.bernoulli(torch.full(shape, 1.0 - p))
    t0 = time.perf_counter()
    loss = None
    for i in range(args.steps):
        ids = bench.markov_sessions(B, L, active, 10 + i)
        bern = torch.rand(B, L) < 0.15
        lens = (ids != 0).sum(1)
        j1 = (torch.rand(B) * lens).long()
        m, lab = O.mlm_targets_train(ids, bern, j1, lambda mm: mm.float().argmax(1))
        opt.zero_grad()
        x = O.apply_mask_mlm(O.embedding_lookup(ids, table), m, memb, True, False)
        if p > 0:
            s = 1.0 / (1.0 - p)
            h = x * keep(B, L, D) * s
            pos_mask = keep(B, 2 * L, D)
            for lp in layers:
                masks = dict(pos=pos_mask, prob=keep(B, n, L, L), attn_out=keep(B, L, D), ff_act=keep(B, L, 4 * D),
                             ff_out=keep(B, L, D))
                h = O.xlnet_layer_dropout(h, lp, n, 0.03, masks, p)
            h = h * keep(B, L, D) * s
        else:
            h = O.xlnet_model(x, layers, n, 0.03)
        xr, y = O.remove_pad_rows(h, lab)
        loss = O.cross_entropy(O.head_logits(xr, table, 1.0), y)
        loss.backward()
        opt.step()
        if i % 20 == 0:
            print(f"step {i}: loss {float(loss):.4f} ({time.perf_counter() - t0:.0f} s)", flush=True)
    train_s = time.perf_counter() - t0
    rec = ndcg = 0.0
    cnt = 0
    with torch.no_grad():
        for j in range(4):
            ids = bench.markov_sessions(B, L, active, 900_000 + j)
            m, lab = O.mlm_targets_eval(ids)
            x = O.apply_mask_mlm(O.embedding_lookup(ids, table), m, memb, False, True)
            h = O.xlnet_model(x, layers, n, 0.03)
            xr, y = O.remove_pad_rows(h, lab)
            logits = O.head_logits(xr, table, 1.0)
            rec += float(O.recall_at_k(logits, y, 20).sum())
            ndcg += float(O.ndcg_at_k(logits, y, 20).sum())
            cnt += y.numel()
    res = {"what": "CPU oracle (oracle/t4r_oracle.py) at the benchmarked configuration, the counterpart of bench.py's "
                   "recall_at_20.hip_bench_config",
           "config": f"V={V + 1} rows, d={D}, {NL} layers x {n} heads, batch {B}, seq {L}, MLM 0.15, dropout {p}, Adam lr 2e-3, "
                     f"{args.steps} steps on bench.markov_sessions (2000 active items), 4 x {B} held-out sessions, last-item protocol",
           "recall_at_20": round(rec / cnt, 4), "ndcg_at_20": round(ndcg / cnt, 4), "final_train_loss": round(float(loss), 4),
           "train_steps": args.steps, "eval_sessions": cnt, "host_threads": torch.get_num_threads(), "host_cores": cores,
           "train_seconds": round(train_s, 1), "sessions_per_s": round(B * args.steps / train_s, 1),
           "measured_where": "build container (no GPU), committed; the HIP value of the same probe is in every bench line"}
    print(json.dumps(res), flush=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
