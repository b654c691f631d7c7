"""TEST / MEASUREMENT INFRASTRUCTURE ONLY.  The CPU oracle trained in LOCKSTEP with the HIP path at the BENCHMARKED
configuration (VERDICT r5 next #1b): BASELINE configs[1] -- 100 001 x 128 table, 4-layer 4-head XLNet, batch 1024, seq 20,
MLM p = 0.15, **dropout 0.3**, tied full softmax, Adam lr 2e-3 -- for `--steps` steps on the Markov-chain sessions of
bench.recall_probe, evaluated on the same four held-out batches.

Lockstep means: the same initial parameters (bench.build_modules(seed)), the same sessions, and the SAME random decisions
at every site of every step -- the MLM targets and the seven kinds of dropout masks are computed here by the integer
restatement of the device streams (oracle/device_rng.py; tests/test_round6_gpu.py checks it bit for bit against masks
exported from the device), from the two Philox keys alone.  Nothing is read from a GPU: this script runs wherever a CPU is,
tools/lockstep_bench_config.py runs the HIP side, and `--hip <its json>` merges the two into one record (per-step loss
difference, final Recall@20 / NDCG@20 of both).  If the HIP path computed anything differently from the oracle at this
setting -- a dropout site, the shared pos_emb mask, the head at dropout-regime activations -- the trajectories would part
from step 0; rounding differences alone stay small until the loss leaves its plateau and are bounded by the spread between
seeds afterwards.

    python oracle/cpu_lockstep.py [--steps 200] [--seed 0] [--threads 8] [--out profiles/r06_lockstep_cpu.json] [--hip FILE]
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
import build_c  # noqa: E402
import device_rng as R  # noqa: E402
import golden_utils as gu  # noqa: E402
import t4r_oracle as O  # noqa: E402


def merge(cpu, hip):
    n = min(len(cpu["loss_per_step"]), len(hip["loss_per_step"]))
    d = [abs(a - b) for a, b in zip(cpu["loss_per_step"][:n], hip["loss_per_step"][:n])]
    first = next((i for i, x in enumerate(d) if x > 1e-3), None)
    return {"what": "HIP path vs CPU oracle in lockstep at the benchmarked configuration (dropout 0.3): same init, sessions, MLM "
                    "targets and dropout masks at every site (oracle/device_rng.py restates the device streams)",
            "config": cpu["config"], "steps_compared": n,
            "init_checksum_equal": cpu["init_checksum"] == hip["init_checksum"],
            "max_abs_loss_diff_first_20_steps": max(d[:20]), "max_abs_loss_diff_first_50_steps": max(d[:50]),
            "max_abs_loss_diff_all": max(d), "mean_abs_loss_diff_all": sum(d) / n,
            "first_step_with_loss_diff_above_1e-3": first,
            "loss_at": {str(i): {"hip": hip["loss_per_step"][i], "cpu": cpu["loss_per_step"][i]}
                        for i in sorted({0, 1, 2, 5, 10, 20, 50, 100, 150, n - 1}) if i < n},
            "final": {"hip": {k: hip[k] for k in ("recall_at_20", "ndcg_at_20", "final_train_loss")},
                      "cpu_oracle": {k: cpu[k] for k in ("recall_at_20", "ndcg_at_20", "final_train_loss")}},
            "hip_measured_on": hip.get("measured_on"), "cpu_host_threads": cpu["host_threads"], "cpu_train_seconds": cpu["train_seconds"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0, help="torch.manual_seed of the initialisation (bench.build_modules)")
    ap.add_argument("--mask-seed", type=int, default=1234)
    ap.add_argument("--drop-seed", type=int, default=4321)
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_lockstep_cpu.json"))
    ap.add_argument("--hip", default=None, help="json written by tools/lockstep_bench_config.py: merge and write <out>.merged")
    ap.add_argument("--merge-only", action="store_true")
    args = ap.parse_args()
    if args.merge_only:
        with open(args.out) as f, open(args.hip) as h:
            res = merge(json.load(f), json.load(h))
        print(json.dumps(res))
        with open(args.out.replace(".json", "_merged.json"), "w") as f:
            json.dump(res, f, indent=1)
        return
    build_c.build()
    assert R.use_c()
    cores = os.cpu_count() or 1
    torch.set_num_threads(args.threads or min(cores, 32))
    V, D, n, NL, B, L, p = bench.V_ITEMS, bench.D_MODEL, bench.N_HEAD, bench.N_LAYER, bench.BATCH, bench.SEQ, args.dropout
    tr, schema, model = bench.build_modules(p, seed=args.seed)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    P = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
    table = P["tables"]["item_id"]
    checksum = [round(float(table.detach().double().abs().sum()), 6), round(float(P["layers"][0]["q"].detach().double().abs().sum()), 9)]
    leaves = [table, P["masked_item_embedding"]] + [t for lp in P["layers"] for t in lp.values()]
    opt = torch.optim.Adam(leaves, lr=2e-3)
    cfg = dict(n_head=n, eps=0.03, item="item_id", masking="mlm")
    active = 1 + torch.arange(2000) * (V // 2000)
    losses = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        ids = bench.markov_sessions(B, L, active, 10 + i)
        mask, labels = R.mlm_targets_train_device(ids, args.mask_seed, i * B * L, 0.15)
        drop = None
        if p > 0:
            drop = (p, R.xlnet_dropout_masks(B, L, D, n, NL, p, seed=args.drop_seed, offset=i + 1))
        opt.zero_grad()
        ref = O.session_forward(P, cfg, {"item_id": ids}, mask, labels, True, False, drop=drop)
        ref["loss"].backward()
        opt.step()
        losses.append(float(ref["loss"].detach()))
        if i % 10 == 0:
            print(f"step {i}: loss {losses[-1]:.6f} ({time.perf_counter() - t0:.0f} s)", flush=True)
    train_s = time.perf_counter() - t0
    rec = ndcg = 0.0
    cnt = 0
    with torch.no_grad():
        for j in range(4):
            ids = bench.markov_sessions(B, L, active, 900_000 + j)
            m, lab = O.mlm_targets_eval(ids)
            ro = O.session_forward(P, cfg, {"item_id": ids}, m, lab, False, True)
            rec += float(O.recall_at_k(ro["logits"], ro["labels"], 20).sum())
            ndcg += float(O.ndcg_at_k(ro["logits"], ro["labels"], 20).sum())
            cnt += ro["labels"].numel()
    res = {"what": "CPU oracle at the benchmarked configuration, every random decision taken from the restated device streams",
           "config": f"V={V + 1} rows, d={D}, {NL} layers x {n} heads, batch {B}, seq {L}, MLM 0.15, dropout {p}, Adam lr 2e-3, "
                     f"{args.steps} steps on bench.markov_sessions (2000 active items), 4 x {B} held-out sessions, last-item protocol; "
                     f"init seed {args.seed}, MLM key {args.mask_seed}, dropout key {args.drop_seed}",
           "init_checksum": checksum, "loss_per_step": [round(x, 6) for x in losses],
           "recall_at_20": round(rec / cnt, 4), "ndcg_at_20": round(ndcg / cnt, 4), "final_train_loss": round(losses[-1], 4),
           "train_steps": args.steps, "eval_sessions": cnt, "host_threads": torch.get_num_threads(), "host_cores": cores,
           "train_seconds": round(train_s, 1)}
    print(json.dumps({k: v for k, v in res.items() if k != "loss_per_step"}), flush=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    if args.hip:
        with open(args.hip) as h:
            mg = merge(res, json.load(h))
        print(json.dumps(mg))
        with open(args.out.replace(".json", "_merged.json"), "w") as f:
            json.dump(mg, f, indent=1)


if __name__ == "__main__":
    main()
