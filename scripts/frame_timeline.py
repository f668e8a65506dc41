#!/usr/bin/env python3
"""Where a frame goes, kernel by kernel, from a rocprofv3 --kernel-trace rocpd database: the launches of a few consecutive steady-state frames in
time order with their durations and the idle gap in front of each (copy / fill commands of the runtime show up as kernels too).
usage: frame_timeline.py kt_results.db [frames=4] [anchor kernel name fragment = k_rm_prepare]"""
import sqlite3, sys
db = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
anchor = sys.argv[3] if len(sys.argv) > 3 else "k_rm_prepare"
con = sqlite3.connect(db)
cur = con.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
src = None
for v in views:
    try:
        cols = [d[1] for d in cur.execute(f"pragma table_info('{v}')")]
    except Exception:
        continue
    if {"start", "end", "name"} <= set(cols) and ("kernel" in v.lower()):
        src = v
        break
if src is None:
    sys.exit(f"no kernel view with start/end/name among {views}")
rows = cur.execute(f"select name, start, end from {src} order by start").fetchall()
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
if len(idx) < frames + 8:
    sys.exit(f"only {len(idx)} launches of {anchor}")
mid = len(idx) // 2
mid -= mid % 2                         # start on an even frame of the run (the DEMO bench refills on even frames)
lo, hi = idx[mid], idx[mid + frames]
# a frame starts with its bin kernels when it refills: walk back to the first launch after the previous frame's k_raymarch
while lo > 0 and "k_raymarch" not in rows[lo - 1][0]:
    lo -= 1
t0 = rows[lo][1]
prev_end = rows[lo - 1][2] if lo > 0 else t0
print(f"{'t_us':>9s} {'gap_us':>7s} {'dur_us':>8s}  kernel      ({src} of {db})")
busy = 0.0
for name, s, e in rows[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} {(s - prev_end) / 1e3:7.1f} {(e - s) / 1e3:8.1f}  {short(name)}")
    busy += (e - s) / 1e3
    prev_end = max(prev_end, e)
span = (rows[hi - 1][2] - t0) / 1e3
print(f"{frames} frames: {span:.1f} us, kernels busy {busy:.1f} us ({100 * busy / span:.0f} %), {span / frames:.1f} us per frame")
