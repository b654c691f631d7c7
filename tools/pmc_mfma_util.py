#!/usr/bin/env python
"""MFMA utilisation per kernel from a rocprofv3 PMC pass (rocpd database):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d DIR -o NAME -- python bench.py --steps 4 --warmup 2 ...
    python tools/pmc_mfma_util.py DIR/.../NAME_results.db > profiles/rNN_pmc_mfma_util.csv
utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): the fraction of the
kernel's active cycles in which a SIMD's matrix pipe is busy (both counters are summed over their
instances by rocprofv3).  Kernels are serialised under the profiler, so in-step overlap is not visible here."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v in rows:
    agg[re.sub(r"^void ", "", k)[:90]][c].append(v)
out = []
for k, d in agg.items():
    act, mf = d.get("GRBM_GUI_ACTIVE", []), d.get("SQ_VALU_MFMA_BUSY_CYCLES", [])
    if not act or not mf or sum(mf) == 0:
        continue
    a, m = sum(act) / len(act), sum(mf) / len(mf)
    out.append((sum(act), k, len(act), a, m, (m / 1024.0) / (a / 8.0)))
print("kernel,dispatches,avg_GRBM_GUI_ACTIVE,avg_SQ_VALU_MFMA_BUSY_CYCLES,mfma_utilisation")
for _, k, n, a, m, u in sorted(out, reverse=True):
    print('"%s",%d,%.0f,%.0f,%.3f' % (k, n, a, m, u))
