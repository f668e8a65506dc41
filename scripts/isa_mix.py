#!/usr/bin/env python3
"""Static VALU opcode mix of the hot kernels from the gfx950 ISA listing (hipcc -S, no GPU needed), loop-weighted: every instruction counts
8^depth, depth = number of loops (backward branches) enclosing it.  Together with the per-opcode issue rates and class membership measured by
scripts/probes/valu_rate_probe.hip under rocprofv3 (profiles/r04_valu_classes.json) this gives, per SQ_INSTS_VALU_* class counter, the average
issue cycles of an instruction of that class IN THIS KERNEL -- what scripts/limiters_json.py multiplies the class counters of a profiled run with.
usage: isa_mix.py profiles/r04_valu_classes.json out.json   (compiles csrc/fill.hip, raymarch.hip and the two *_generic.hip units: ~3 minutes)"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc")
cal = json.load(open(sys.argv[1]))["opcodes"]
RATE = {op.split()[0]: d["cycles_per_wave_instruction_per_simd"] for op, d in cal.items() if " " not in op.strip() or op.startswith("v_cndmask_b32 (")}
CLS = {op.split()[0]: d["class"] for op, d in cal.items()}
RATE["v_cndmask_b32"] = cal.get("v_cndmask_b32 (sgpr mask)", {}).get("cycles_per_wave_instruction_per_simd", 4.26)   # (the vcc form of the probe measures a hazard, not the rate)
FULL, HALF, TRANS = 2.6, 4.3, 8.25
def classify(op):
    """(class counter, issue cycles per wave-instruction per SIMD) of a VALU opcode; measured where the probe has it, else by family"""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in RATE: return CLS.get(base, "OTHER"), RATE[base]
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", base): return "TRANS_F32", TRANS
    if base.startswith("v_cvt_"): return "CVT", HALF
    if base.startswith("v_pk_"): return ("FMA_F32" if "fma" in base else "MUL_F32" if "mul" in base else "ADD_F32" if ("add" in base or "sub" in base) else "OTHER"), HALF
    if re.match(r"v_(fma|fmac|mac|mad)_f32", base): return "FMA_F32", FULL
    if re.match(r"v_fma_mix", base): return "FMA_F32", HALF
    if re.match(r"v_mul_f32", base): return "MUL_F32", FULL
    if re.match(r"v_(add|sub|subrev)_f32", base): return "ADD_F32", FULL
    if re.match(r"v_(add|sub|subrev)(_co)?_u32", base) or re.match(r"v_(addc|subb)_co_u32", base): return "INT32", FULL
    if re.match(r"v_(max|min)_[iu]32|v_mul_[iu]32_[iu]24|v_mad_[iu]32_[iu]24|v_mul_lo_u32|v_mul_hi_u32|v_lshl_add_u32|v_add3_u32|v_lshl_or_b32|v_add_lshl_u32", base): return "INT32", HALF
    if re.match(r"v_mov_b32|v_lshlrev_b32|v_and_b32", base): return "OTHER", 2.72 if not base.startswith("v_mov") else 2.25
    return "OTHER", HALF
def listing(name):
    d = tempfile.mkdtemp()
    asm = os.path.join(d, name + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S",
                    os.path.join(CSRC, name + ".hip"), "-o", asm], check=True, capture_output=True)
    return open(asm).read()
def functions(txt):
    cur, body = None, []
    for line in txt.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m: cur, body = m.group(1), []; continue
        if cur and line.startswith(".Lfunc_end"):
            yield cur, body; cur = None; continue
        if cur is not None: body.append(line)
def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))
def mix(body):
    labels, insts = {}, []
    for line in body:
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m: labels[m.group(1)] = len(insts); continue
        m = re.match(r"^\t([a-z_0-9]+)\s*(.*)", line)
        if m and not m.group(1).startswith("."): insts.append((m.group(1), m.group(2)))
    depth = [0] * (len(insts) + 1)
    for i, (op, args) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            t = labels.get(args.split()[0].strip())
            if t is not None and t <= i:
                for j in range(t, i + 1): depth[j] += 1
    agg = {}
    for i, (op, _) in enumerate(insts):
        if not op.startswith("v_") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") and False: continue
        c, r = classify(op)
        w = 8.0 ** depth[i]
        a = agg.setdefault(c, {"w": 0.0, "wr": 0.0, "wi": 0.0, "ops": {}})
        a["w"] += w; a["wr"] += w * r
        a["wi"] += w * (2.0 if r < 3.2 else 8.0 if 6.0 < r < 12.0 else 4.0)      # the pipe's nominal rate: SIMD-32 full rate 2, half rate 4, transcendental 8 cycles per wave64
        a["ops"][op] = a["ops"].get(op, 0) + 1
    return {c: {"avg_issue_cycles": a["wr"] / a["w"], "avg_nominal_cycles": a["wi"] / a["w"], "static_ops": dict(sorted(a["ops"].items(), key=lambda kv: -kv[1])[:12])} for c, a in agg.items()}
import hashlib
def kernel_sources_sha():
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)):
        if fn in ("fill.hip", "fill_generic.hip", "fill_kernels.h", "raymarch.hip", "raymarch_generic.hip", "raymarch_kernels.h", "bin.hip", "vpfx_internal.h"):
            h.update(fn.encode() + b"\0" + open(os.path.join(CSRC, fn), "rb").read())
    return h.hexdigest()[:16]
out = {"method": __doc__.split("usage")[0].strip(), "kernel_sources_sha": kernel_sources_sha(), "kernels": {}}
for src in ("fill", "raymarch", "fill_generic", "raymarch_generic"):
    fns = dict(functions(listing(src)))
    dm = demangle(list(fns))
    for mangled, body in fns.items():
        name = dm[mangled].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        # the headline instantiations + (round 6) the cliff / any-nv paths profiled by scripts/gpu_prof_r6.sh: RGBA16F bricks (GREY = false), run-time voxel count
        if re.match(r"k_fill_lds<(32|64), 0, 1, false(, false)?>|k_raymarch<(32|64), false, false, false, (true|false)>|k_fill<(32|64), 0, 0, true(, false)?>|"
                    r"k_fill_lds<(32|64), 0, 2, false, true>|k_fill<(32|64), 0, 0, true, true>|k_raymarch<0, false, false, false, (true|false)>", name):
            out["kernels"][name] = mix(body)
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["kernels"].items(): print(k, {c: round(d["avg_issue_cycles"], 2) for c, d in v.items()})
