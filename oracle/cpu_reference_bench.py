#!/usr/bin/env python
"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by the product path.

CPU timing of the UNMODIFIED reference (NVIDIA-Merlin/Transformers4Rec, imported from /root/reference
through oracle/ref_standins.py) on the benchmark workload of bench.py: BASELINE.json configs[1] --
item vocabulary 100 k (100 001 table rows), d_model 128, 4-layer 4-head XLNet, seq_len 20, MLM p = 0.15,
tied-weight full softmax, dropout 0.3 (XLNetConfig.build default), Adam, one step = forward + backward +
optimizer over one synthetic batch -- i.e. the reference's own plain loop (torch/model/base.py:669-718).

/root/reference exists only in the build container, so this cannot run on the GPU box: the number it
prints is recorded under profiles/ together with the host it ran on (core count stated), next to the
`cpu_baseline` leg of bench.py, which times the oracle port on the GPU box's own host cores.

    python oracle/cpu_reference_bench.py [--batch 1024] [--steps 6] [--threads N]
"""
import argparse
import json
import os
import platform
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import ref_standins as rs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--out", default=None, help="also write the record here (profiles/r06_cpu_reference_bench.json)")
    args = ap.parse_args()
    if not os.path.isdir(rs.REFERENCE_ROOT):
        raise SystemExit(f"{rs.REFERENCE_ROOT} not present: the reference-verbatim timing runs in the build container only")
    torch.set_num_threads(args.threads)
    tr = rs.import_reference()
    import make_golden as mg
    from transformers4rec.config import transformer as tconf

    V, L, D, NH, NL = 100_000, 20, 128, 4, 4
    schema = mg.make_schema(V, L)
    torch.manual_seed(0)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm",
                                                    embedding_dim_default=D)
    cfg = tconf.XLNetConfig.build(d_model=D, n_head=NH, n_layer=NL, total_seq_length=L, dropout=args.dropout)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    import transformers4rec_amd as hip          # only for the Schema-driven synthetic batches (host code)

    hschema = hip.session_schema(V, L)
    times = []
    for i in range(args.warmup + args.steps):
        x = hip.random_data_from_schema(hschema, args.batch, L, seed=100 + i)
        t0 = time.perf_counter()
        opt.zero_grad()
        out = model(x, training=True)
        out["loss"].backward()
        opt.step()
        dt = time.perf_counter() - t0
        if i >= args.warmup:
            times.append(dt)
        print(f"step {i}: {dt:.3f} s  loss {float(out['loss']):.4f}", flush=True)
    times.sort()
    med = times[len(times) // 2]
    res = {"what": "reference-verbatim CPU training step (transformers4rec.torch unmodified, HF XLNetModel)",
           "workload": "BASELINE.json configs[1]: V=100k, d=128, 4x4 XLNet, L=20, MLM, tied full softmax, "
                       f"dropout {args.dropout}, Adam, batch {args.batch}",
           "threads": args.threads, "host_cores": os.cpu_count(), "host": platform.processor() or platform.machine(),
           "steps_timed": len(times), "s_per_step_median": round(med, 4), "s_per_step_min": round(times[0], 4),
           "sessions_per_s_median": round(args.batch / med, 1), "sessions_per_s_best": round(args.batch / times[0], 1),
           "sessions_per_s_worst": round(args.batch / times[-1], 1), "s_per_step_all": [round(t, 4) for t in times],
           "date": time.strftime("%Y-%m-%d"), "torch": torch.__version__,
           "note": "a Firecracker microVM with 8 vCPUs: identical runs of this script have given 67 to 416 sessions/s on different "
                   "days (VERDICT r5); quote min / median / n, never a single figure"}
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
