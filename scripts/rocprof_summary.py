#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel stats and per-kernel PMC counter means."""
import glob, sqlite3, sys
def q(db, sql):
    con = sqlite3.connect(db); cur = con.cursor(); cur.execute(sql)
    cols = [d[0] for d in cur.description]; rows = cur.fetchall(); con.close(); return cols, rows
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
for db in sys.argv[1:]:
    print(f"== {db}")
    try:
        _, rows = q(db, "select name,total_calls,total_duration,average,percentage from top_kernels")
        print(f"{'kernel':48s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}")
        for n, c, t, a, p in rows: print(f"{short(n):48s} {c:6d} {t:12.1f} {a:12.1f} {p:7.2f}")
    except Exception as e: print("no top_kernels:", e)
    try:
        _, rows = q(db, "select kernel_name, counter_name, avg(value), count(*), avg(duration), max(vgpr_count), max(sgpr_count), max(lds_block_size), max(scratch_size) from counters_collection group by kernel_name, counter_name")
        if rows:
            print(f"{'kernel':40s} {'counter':24s} {'mean/dispatch':>18s} {'n':>4s} {'avg_ns':>12s} vgpr sgpr lds scratch")
            for n, cn, v, k, d, vg, sg, lds, sc in rows: print(f"{short(n):40s} {cn:24s} {v:18.1f} {k:4d} {d:12.0f} {vg} {sg} {lds} {sc}")
    except Exception as e: print("no counters:", e)

# optional: --traffic-json OUT : HBM traffic per launch of the two dominant kernels from the FETCH_SIZE / WRITE_SIZE passes
