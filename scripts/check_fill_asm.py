#!/usr/bin/env python3
"""Static check of the software-pipelined inline-asm loads in k_fill (csrc/fill.hip) and k_raymarch (csrc/raymarch.hip).

The loads are inline asm (`global_load_dwordx4 vDST, vOFF, s[..]`) whose asynchronous register write the compiler cannot see;
correctness needs that NO instruction reads or writes vDST between the load and the `s_waitcnt vmcnt(N)` that retires it
(loads return in order, so a wait with N younger loads outstanding retires it).  The compiler is free to insert register copies,
so this script re-derives the property from the generated ISA of every k_fill instantiation and fails loudly if it is violated.
The same holds for the explicitly issued texel loads of k_raymarch's two-sample loop (vaddr form); there every 16-byte global
load of the kernel is tracked (the compiler's own loads satisfy the property by construction).
`fill_lds` checks the LDS-resident cube-map kernels (k_fill_lds) the same way: the loads are `ds_read_u8 vDST, vADDR`, retired by
`s_waitcnt lgkmcnt(N)` (LDS operations return in order among themselves; the compiler's scalar loads share the counter and can
only make a wait stricter).
usage: check_fill_asm.py [fill|fill_lds|raymarch] [extra hipcc flags]      (compiles to asm with hipcc; exit code 1 on violation)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WHICH = sys.argv[1] if len(sys.argv) > 1 else "fill"
CSRC = os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc")
LDS = WHICH == "fill_lds"
SRC = WHICH if WHICH.endswith(".hip") else os.path.join(CSRC, ("fill" if LDS else WHICH) + ".hip")
KERNEL = "k_raymarch" if "raymarch" in os.path.basename(SRC) else ("k_fill_lds" if LDS else "k_fill")
LOAD_RE = (r"global_load_dwordx[24]\s+(v\[\d+:\d+\]),\s*v\[\d+:\d+\],\s*off" if KERNEL == "k_raymarch"
           else r"ds_read_u8\s+(v\d+),\s*v\d+" if LDS
           else r"global_load_dwordx4\s+(v\[\d+:\d+\]),\s*v\d+,\s*s\[\d+:\d+\]")
COUNTER = "lgkmcnt" if LDS else "vmcnt"
EXTRA = sys.argv[2:]


def regs(tok):
    """'v[8:11]' / 'v12' -> set of VGPR numbers"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def pk_touched(t):
    """Registers a packed-f32 instruction really consumes: a 64-bit source operand whose op_sel / op_sel_hi select the same half
    for both lanes (e.g. a scalar weight broadcast with op_sel_hi:[0,..]) only uses that one register; the other one is not an
    input, so a pending load into it is no hazard."""
    head = re.split(r"\s+(?:op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b", t)[0]
    ops = [o.strip() for o in head.split(None, 1)[1].split(",")]
    sel = [0, 0, 0]
    sel_hi = [1, 1, 1]
    m = re.search(r"op_sel:\[([\d,]+)\]", t)
    if m:
        sel = [int(x) for x in m.group(1).split(",")] + [0, 0, 0]
    m = re.search(r"op_sel_hi:\[([\d,]+)\]", t)
    if m:
        sel_hi = [int(x) for x in m.group(1).split(",")] + [1, 1, 1]
    out = set(regs(ops[0]))                                  # destination pair
    for k, o in enumerate(ops[1:4]):
        mm = re.match(r"v\[(\d+):(\d+)\]$", o)
        if not mm:
            out |= regs(o)
            continue
        lo, hi = int(mm.group(1)), int(mm.group(2))
        if sel[k] == 0 or sel_hi[k] == 0:
            out.add(lo)
        if sel[k] == 1 or sel_hi[k] == 1:
            out.add(hi)
    return out


def check_kernel(lines):
    outstanding = []            # [(dest regs, line no)] in issue order; only the asm saddr-form loads are tracked
    bad = []
    for no, l in lines:
        t = l.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        op = t.split()[0]
        m = re.match(LOAD_RE, t)
        if m:
            outstanding.append((regs(m.group(1)), no))
            continue
        if op == "s_waitcnt":
            mm = re.search(COUNTER + r"\((\d+)\)", t)
            if mm:
                n = int(mm.group(1))
                outstanding = [] if n == 0 else outstanding[-n:]
            continue
        # younger compiler-issued VMEM ops only make our waits stricter (in-order return): safe to ignore
        touched = pk_touched(t) if op.startswith("v_pk_") else regs(t)
        for dest, lno in outstanding:
            if touched & dest:
                bad.append((no, t, lno))
    return bad


def main():
    pre = os.environ.get("VPFX_ASM_FILE")           # tests compile each source once and hand the listing to every check of it
    if pre and os.path.exists(pre) and not EXTRA:
        txt = open(pre).read().split("\n")
    else:
      with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "fill.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only",
               "-I", os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc"), "-S", SRC, "-o", out] + EXTRA
        subprocess.run(cmd, check=True, capture_output=True)
        txt = open(out).read().split("\n")
    kernels, cur = {}, None
    for i, l in enumerate(txt):
        m = re.match(r"^(_ZN\S*" + KERNEL + r"\S*):", l)
        if m:
            cur = []
            kernels[m.group(1)] = cur
        elif cur is not None:
            cur.append((i + 1, l))
            if "s_endpgm" in l:
                cur = None
    nbad, nloads = 0, 0
    for name, lines in kernels.items():
        nloads += sum(1 for _, l in lines if re.search(LOAD_RE, l))
        for no, t, lno in check_kernel(lines):
            print(f"VIOLATION in {name[:70]}: line {no}: '{t}' touches the destination of the load issued at line {lno}")
            nbad += 1
    print(f"checked {len(kernels)} {KERNEL} instantiations, {nloads} pipelined loads, {nbad} violations")
    return 1 if nbad or not nloads else 0


if __name__ == "__main__":
    sys.exit(main())
