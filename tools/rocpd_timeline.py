#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 rocpd database (kernel-trace): every dispatch of the step in start
order with its queue, start offset, duration and the gap to the previous dispatch on the same queue, plus the step's
wall time, the union of busy intervals (any queue) and the per-queue busy time -- shows where the device idles.
    python tools/rocpd_timeline.py DB [--step K] [--anchor SUBSTR]
A step starts at a dispatch whose kernel name contains the anchor (default: mask_targets_kernel)."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = (f"select s.{name_col}, d.start, d.end, {('d.' + qcol) if qcol else '0'} from rocpd_kernel_dispatch d "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start")
    rows = list(cur.execute(q))
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "mask_targets_kernel"
    k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -3
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(starts) < 4:
        print("too few steps", len(starts))
        return
    i0, i1 = starts[k], starts[k + 1]
    step = rows[i0:i1]
    t0 = step[0][1]
    wall = rows[i1][1] - t0
    iv = sorted((s, e) for _, s, e, _ in step)
    busy, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    per_q = {}
    for n, s, e, qq in step:
        per_q[qq] = per_q.get(qq, 0) + e - s
    print(f"# step wall {wall / 1e3:.1f} us, union busy {busy / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us, "
          f"sum of kernel time {sum(e - s for _, s, e, _ in step) / 1e3:.1f} us, dispatches {len(step)}")
    print("# per-queue busy us:", {qq: round(v / 1e3, 1) for qq, v in per_q.items()})
    last_end = {}
    print("start_us,dur_us,gap_same_queue_us,queue,name")
    for n, s, e, qq in step:
        n = re.sub(r"\[clone.*?\]", "", n)
        n = re.sub(r"^void\s+", "", n)[:70]
        gap = (s - last_end[qq]) / 1e3 if qq in last_end else 0.0
        last_end[qq] = max(last_end.get(qq, 0), e)
        print(f"{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f},{qq},{n}")


if __name__ == "__main__":
    main()
