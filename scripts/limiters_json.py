#!/usr/bin/env python3
"""What limits the two dominant kernels instead of HBM, as NUMBERS derived from rocprofv3 PMC passes (one rocpd .db per pass), written to
profiles/limiters_<cfg>_<cubemap>.json under the same kernel-source fingerprint as the traffic file; bench.py reports them as
`roofline.limiter` only while the fingerprint still matches.  usage: limiters_json.py out.json pass1.db pass2.db ..."""
import hashlib, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn in ("fill.hip", "raymarch.hip", "bin.hip", "vpfx_internal.h"):
            h.update(fn.encode() + b"\0" + open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]
ctr = {}
for db in sys.argv[2:]:
    con = sqlite3.connect(db); cur = con.cursor()
    try:
        cur.execute("select kernel_name, counter_name, avg(value), avg(duration) from counters_collection group by kernel_name, counter_name")
        for n, c, v, d in cur.fetchall():
            k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
            full = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if k in ("k_fill", "k_fill_lds", "k_raymarch") and "k_raymarch_one" not in full:
                ctr.setdefault(k, {"instantiation": full})[c] = v
                ctr[k].setdefault("avg_ns_under_pmc", d)
    finally:
        con.close()
SIMDS = 1024
out = {"kernel_sources_sha": kernel_sources_sha(), "source": "rocprofv3 --pmc passes of `python bench.py --steps 10 --warmup 2 --no-cpu-baseline` (scripts/gpu_prof_r3.sh)"}
for k, c in ctr.items():
    g = lambda name: c.get(name)
    d = {"instantiation": c["instantiation"]}
    cyc = g("GRBM_GUI_ACTIVE")
    if cyc: cyc /= 8.0                          # the counter is summed over the 8 XCDs
    if g("SQ_INSTS_VALU") is not None: d["valu_wave_instructions_per_launch"] = g("SQ_INSTS_VALU")
    if g("SQ_INSTS_VALU") is not None and g("SQ_INSTS_VMEM_RD"): d["valu_per_vmem_read"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_VMEM_RD")
    if g("SQ_ACTIVE_INST_VALU") is not None and cyc: d["valu_issue_busy_fraction"] = g("SQ_ACTIVE_INST_VALU") * 4.0 / (SIMDS * cyc)   # ~1 = VALU-issue-bound
    if g("SQ_THREAD_CYCLES_VALU") is not None and g("SQ_ACTIVE_INST_VALU"): d["lanes_enabled_per_valu_fraction"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"): d["lds_conflict_share_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_LDS_IDX_ACTIVE") is not None and cyc: d["lds_pipe_busy_fraction"] = g("SQ_LDS_IDX_ACTIVE") / (256.0 * cyc)
    if g("SQ_INSTS_LDS") is not None: d["lds_wave_instructions_per_launch"] = g("SQ_INSTS_LDS")
    if g("SQ_INSTS_VMEM_RD") is not None: d["vmem_read_wave_instructions_per_launch"] = g("SQ_INSTS_VMEM_RD")
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") is not None and cyc: d["l1_accesses_per_cu_cycle"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") / (256.0 * cyc)
    if g("SQ_WAVE_CYCLES") is not None and cyc: d["wave_cycles_x4_per_simd_cycle"] = g("SQ_WAVE_CYCLES") * 4.0 / (SIMDS * cyc)   # average waves resident per SIMD
    if cyc: d["gpu_cycles_per_launch"] = cyc
    d["limited_by"] = "VALU issue" if d.get("valu_issue_busy_fraction", 0) >= 0.7 else "see the fractions"
    out[k] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
