#!/usr/bin/env python3
"""Which SQ_INSTS_VALU_* class counter does each VALU opcode land in on gfx950, and what does SQ_ACTIVE_INST_VALU read on a SIMD whose VALU is
saturated?  Input: rocpd .db files of `rocprofv3 --pmc ... -- scripts/probes/_bin/valu_rate_probe` (scripts/gpu_valu_class.sh); every dispatch of the
probe is one opcode at a known instruction count, so counter / instructions is the class membership (0 or 1) and the issue-side counters can be
normalised per SIMD-cycle.  Output: a table (profiles/r04_valu_class_calibration.txt) and the JSON that scripts/limiters_json.py reads.
usage: valu_class_calibration.py probe_stdout.txt out.json pass1.db pass2.db ..."""
import json, re, sqlite3, sys
stdout, out_json, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
# the probe prints one line per dispatch, in dispatch order: name, ILP, waves/SIMD, exec mask, cycles per wave-instruction per SIMD
lines = [l for l in open(stdout) if "cycles per wave-instruction" in l]
runs = []
for l in lines:
    m = re.match(r"\s*(.+?)\s+ILP (\d)\s+waves/SIMD (\d)\s+exec ([0-9a-f]+) :\s+([0-9.]+)", l)
    runs.append({"op": m.group(1), "ilp": int(m.group(2)), "wps": int(m.group(3)), "mask": m.group(4), "cyc": float(m.group(5))})
ITERS, UNROLL, NBLK = 2000, 16, 64
per = {}
for db in dbs:
    con = sqlite3.connect(db); cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    key = "dispatch_id" if "dispatch_id" in cols else "id"
    cur.execute(f"select {key}, counter_name, sum(value) from counters_collection where kernel_name like '%k_probe%' group by {key}, counter_name order by {key}")
    rows = cur.fetchall(); con.close()
    ids = sorted({r[0] for r in rows})
    for i, cn, v in rows:
        per.setdefault(ids.index(i), {})[cn] = v
table, classes = [], {}
for i, r in enumerate(runs):
    c = per.get(i, {})
    n_inst = NBLK * 4 * r["wps"] * ITERS * UNROLL * r["ilp"]          # wave-instructions of the probed opcode in this dispatch
    row = dict(r, instructions=n_inst)
    for cn, v in sorted(c.items()):
        row[cn] = v / n_inst if cn.startswith("SQ_INSTS_VALU") else v
    table.append(row)
    if r["ilp"] == 4 and r["wps"] == 4 and r["mask"] == "ffffffffffffffff" and not r["op"].startswith("ds_"):
        member = [cn.replace("SQ_INSTS_VALU_", "") for cn, v in c.items() if cn.startswith("SQ_INSTS_VALU_") and v / n_inst > 0.5]
        classes[r["op"]] = {"class": member[0] if member else "OTHER", "cycles_per_wave_instruction_per_simd": r["cyc"]}
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
            classes[r["op"]]["SQ_ACTIVE_INST_VALU_per_instruction"] = c["SQ_ACTIVE_INST_VALU"] / n_inst
rates = {}
for op, d in classes.items():
    rates.setdefault(d["class"], []).append(d["cycles_per_wave_instruction_per_simd"])
summary = {k: {"min": min(v), "max": max(v), "n_opcodes": len(v)} for k, v in rates.items()}
json.dump({"opcodes": classes, "class_rates_cycles_per_wave_instruction_per_simd": summary}, open(out_json, "w"), indent=1)
print(f"{'opcode':28s} {'class':10s} {'cyc/inst/SIMD':>14s} {'ACTIVE_INST_VALU/inst':>22s}")
for op, d in classes.items():
    print(f"{op:28s} {d['class']:10s} {d['cycles_per_wave_instruction_per_simd']:14.2f} {d.get('SQ_ACTIVE_INST_VALU_per_instruction', float('nan')):22.3f}")
print(json.dumps(summary, indent=1))
