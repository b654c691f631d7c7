import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import bench
for th in (16, 32, 64, 128):
    os.environ["T4R_CPU_BASELINE_THREADS"] = str(th)
    t0 = time.time(); r = bench.cpu_baseline(8.0); print(th, r["value"], r["sample"][:20], round(time.time() - t0, 1), flush=True)
