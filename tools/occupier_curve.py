#!/usr/bin/env python
"""Does the training step survive a co-resident collective?  (measured on ONE GPU.)

At N > 1 the 51 MB table all-reduce is launched right after the head's backward and runs UNDER the transformer body's
backward (bench.setup_data_parallel): an RCCL ring kernel keeps one workgroup per channel resident on a CU for
the whole collective.  The body's token-tile kernels launch exactly one 512-thread / ~140 KB-LDS workgroup per CU (256 of
them), so a CU that an RCCL workgroup holds may not be able to take one -- then every body kernel runs in two waves of
workgroups.  This tool reproduces that occupancy with tools/t4r_tools.hip's occupier (k do-nothing workgroups that hold
their CUs from the head's backward to the end of the backward pass) and prints ms/step against k for two footprints:
"channel" (256 threads, 16 KB LDS: what an RCCL channel looks like) and "exclusive" (512 threads, 96 KB LDS: a workgroup
no body workgroup can share a CU with).

    python tools/occupier_curve.py [--steps 60] [--out gpurun_out/r05_occupier_curve.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import t4r_tools  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--priority", type=int, default=0, help="priority of the occupier's stream (-1 = high)")
    ap.add_argument("--only", type=int, default=0, help="one point only: k workgroups of the channel footprint (for a kernel trace)")
    ap.add_argument("--budget", action="store_true",
                    help="tell the backward's token-tile kernels how many CUs are left while the occupier is resident "
                         "(ops.xlnet_set_cu_budget(256 - k): what distributed.GradReducer does around the table all-reduce)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_occupier_curve.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    tr, schema, model, dense, tables, opt = bench.build(dev, 0.3, config=args.config)
    reducer, _ = bench.setup_data_parallel(tr, model, dense, tables, 1)
    batches = [tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=i, device=dev) for i in range(8)]
    model.train()
    state = {"occ": None}
    from transformers4rec_amd import ops

    def on_head_backward():
        occ = state["occ"]
        if occ is not None:
            occ.start()
            if args.budget:
                ops.xlnet_set_cu_budget(256 - occ.k)

    hook = tr.head_backward_hook(model, on_head_backward)

    def step(i):
        out = model(batches[i % 8], training=True)
        out["loss"].backward()
        occ = state["occ"]
        if occ is not None:
            occ.stop()          # in stream order: after the last kernel of the backward pass on the caller's stream
            occ.join()
            ops.xlnet_set_cu_budget(0)
        reducer.reduce_all()
        opt.step(grad_scale=reducer.grad_scale)
        return out

    def timed():
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / args.steps

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 4.0:       # clock ramp
        step(0)
    res = {"what": "ms per training step of bench.py's workload with k resident do-nothing workgroups held from the head's backward "
                   "to the end of the backward pass (one GPU; the occupancy an RCCL all-reduce of the table bucket would add)",
           "config": bench.WORKLOADS[args.config], "steps": args.steps, "curves": {}}
    base = timed()
    res["baseline_ms"] = round(base, 4)
    # ONE side stream for every point of the curve: HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default
    # 4) round-robin, and a stream that lands on the queue of the caller's stream or of a weight-gradient stream is
    # SERIALISED with it -- a resident kernel there blocks everything queued behind it (round 5's first curve, one fresh
    # stream per point, was a lottery: 1.06x at k = 16 and 8x at the next point).  The same holds for RCCL's stream at N > 1.
    side = torch.cuda.Stream(device=dev, priority=args.priority)
    res["gpu_max_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "default (4)")
    res["side_stream_priority"] = args.priority
    res["cu_budget_told"] = bool(args.budget)
    for name, threads, lds in (("channel_256thr_16KB", 256, 16 * 1024), ("exclusive_512thr_96KB", 512, 96 * 1024)):
        if args.only and threads != 256:
            continue
        curve = {}
        for k in ((args.only,) if args.only else (8, 16, 32, 64)):
            state["occ"] = t4r_tools.Occupier(k, threads=threads, lds_bytes=lds, max_us=20000, device=dev)
            state["occ"].side = side
            ms = timed()
            seen = int(state["occ"].seen.item())
            curve[str(k)] = {"ms_per_step": round(ms, 4), "slowdown": round(ms / base, 4),
                             "occupier_workgroups_run": seen, "expected": k * (args.warmup + args.steps)}
            state["occ"] = None
        res["curves"][name] = curve
    res["baseline_ms_after"] = round(timed(), 4)
    hook.remove()
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
