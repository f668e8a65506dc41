#!/usr/bin/env python3
"""What limits the two dominant kernels instead of HBM, as NUMBERS derived from rocprofv3 PMC passes (one rocpd .db per pass), written to
profiles/limiters_<cfg>_<cubemap>.json under the same kernel-source fingerprint as the traffic file; bench.py reports them as
`roofline.limiter` only while the fingerprint still matches.  usage: limiters_json.py out.json pass1.db pass2.db ...
or: limiters_json.py out.json earlier_limiters.json   (recompute the derived figures from the raw counter means an earlier run of this script kept,
e.g. once profiles/r05_isa_mix.json has been regenerated for the sources that run profiled; the fingerprint stays the one recorded there)"""
import hashlib, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn in ("fill.hip", "fill_generic.hip", "fill_kernels.h", "raymarch.hip", "raymarch_generic.hip", "raymarch_kernels.h", "bin.hip", "vpfx_internal.h"):
            h.update(fn.encode() + b"\0" + open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]
ctr = {}
RECORDED_SHA = None
if len(sys.argv) == 3 and sys.argv[2].endswith(".json"):
    _old = json.load(open(sys.argv[2]))
    RECORDED_SHA = _old.get("kernel_sources_sha")
    for _k, _d in _old.items():
        if isinstance(_d, dict) and "raw_counter_means_per_launch" in _d:
            ctr[_k] = dict(_d["raw_counter_means_per_launch"], instantiation=_d["instantiation"])
for db in ([] if RECORDED_SHA else sys.argv[2:]):
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
try:
    _p = os.path.join(ROOT, "profiles", "r06_isa_mix.json")
    _m = json.load(open(_p if os.path.exists(_p) else os.path.join(ROOT, "profiles", "r05_isa_mix.json")))
    ISA_MIX = _m["kernels"] if _m.get("kernel_sources_sha") in (None, kernel_sources_sha()) else {}
    if not ISA_MIX: print("profiles/r0N_isa_mix.json was made on other kernel sources: no valu_issue fraction (rerun scripts/isa_mix.py)", file=sys.stderr)
except Exception:
    ISA_MIX = {}
if RECORDED_SHA and RECORDED_SHA != kernel_sources_sha():
    sys.exit(f"the counters were measured on kernel sources {RECORDED_SHA}, the tree is at {kernel_sources_sha()}: not re-deriving across sources")
out = {"kernel_sources_sha": kernel_sources_sha(), "source": "rocprofv3 --pmc passes of the bench command (scripts/gpu_prof_r5.sh)"}
for k, c in ctr.items():
    g = lambda name: c.get(name)
    d = {"instantiation": c["instantiation"]}
    cyc = g("GRBM_GUI_ACTIVE")
    if cyc: cyc /= 8.0                          # the counter is summed over the 8 XCDs
    if g("SQ_INSTS_VALU") is not None: d["valu_wave_instructions_per_launch"] = g("SQ_INSTS_VALU")
    if g("SQ_INSTS_VALU") is not None and g("SQ_INSTS_VMEM_RD"): d["valu_per_vmem_read"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_VMEM_RD")
    # VALU issue: SQ_ACTIVE_INST_VALU is NOT a busy time on gfx950 -- under the opcode probe it reads exactly 1 quad-cycle per instruction (2 for a
    # transcendental) whatever the opcode's real issue rate (2.2-2.7 / 4.3 / 8.25 cycles per wave64 instruction per SIMD; scripts/gpu_valu_class.sh,
    # profiles/r04_valu_class_calibration.txt), so round 3's `x 4 / SIMD-cycles` exceeded 1.  The fraction reported here prices the class
    # counters of this run (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _INT32, _CVT; the rest = SQ_INSTS_VALU minus those) with the average issue
    # cycles of that class IN THIS KERNEL (loop-weighted static opcode mix of its ISA x the measured per-opcode rates: profiles/r05_isa_mix.json).
    mix = ISA_MIX.get(c["instantiation"])
    cls = {n: g("SQ_INSTS_VALU_" + n) for n in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "CVT")}
    if mix and g("SQ_INSTS_VALU") is not None and all(v is not None for v in cls.values()) and cyc:
        cls["OTHER"] = max(g("SQ_INSTS_VALU") - sum(cls.values()), 0.0)
        need_probe = sum(n * mix.get(k, {}).get("avg_issue_cycles", 4.3) for k, n in cls.items())
        # `frac` prices every instruction at its pipe's NOMINAL rate (2 / 4 / 8 cycles per wave64 instruction on the SIMD-32: nothing issues faster), so it
        # is <= 1 by construction; `frac_probe_rates` uses the rates the single-opcode probe reached (2.2-2.9 / 4.3 / 8.25: ~10-40 % above
        # nominal for the full-rate ops) -- a mixed instruction stream can issue a little faster than those, so this one may pass 1
        need = sum(n * mix.get(k, {}).get("avg_nominal_cycles", 4.0) for k, n in cls.items())
        d["valu_issue"] = {"bound": "valu_issue", "unit": "SIMD issue cycles per launch", "achieved": need, "peak": SIMDS * cyc, "frac": need / (SIMDS * cyc),
                           "frac_probe_rates": need_probe / (SIMDS * cyc),
                           "wave_instructions_by_class": cls,
                           "avg_issue_cycles_by_class": {k: round(mix.get(k, {}).get("avg_issue_cycles", 4.3), 3) for k in cls},
                           "source": "class counters of this profile x profiles/r06_isa_mix.json (opcode rates: profiles/r04_valu_classes.json)"}
    if g("SQ_THREAD_CYCLES_VALU") is not None and g("SQ_ACTIVE_INST_VALU"): d["lanes_enabled_per_valu_fraction"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"): d["lds_conflict_share_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_LDS_IDX_ACTIVE") is not None and cyc: d["lds_pipe_busy_fraction"] = g("SQ_LDS_IDX_ACTIVE") / (256.0 * cyc)
    if g("SQ_INSTS_LDS") is not None: d["lds_wave_instructions_per_launch"] = g("SQ_INSTS_LDS")
    if g("SQ_INSTS_VMEM_RD") is not None: d["vmem_read_wave_instructions_per_launch"] = g("SQ_INSTS_VMEM_RD")
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") is not None and cyc: d["l1_accesses_per_cu_cycle"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") / (256.0 * cyc)
    if g("SQ_WAVE_CYCLES") is not None and cyc: d["wave_cycles_x4_per_simd_cycle"] = g("SQ_WAVE_CYCLES") * 4.0 / (SIMDS * cyc)   # average waves resident per SIMD
    if cyc: d["gpu_cycles_per_launch"] = cyc
    d["raw_counter_means_per_launch"] = {n: v for n, v in c.items() if n != "instantiation"}
    d["limited_by"] = "VALU issue" if d.get("valu_issue", {}).get("frac_probe_rates", 0) >= 0.7 else "see the fractions"
    out[k] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
