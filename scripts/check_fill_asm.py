#!/usr/bin/env python3
"""Static check of the software-pipelined inline-asm loads in k_fill (csrc/fill.hip) and k_raymarch (csrc/raymarch.hip).

The loads are inline asm (`global_load_dwordx4 vDST, vOFF, s[..]`) whose asynchronous register write the compiler cannot see;
correctness needs that NO instruction reads or writes vDST between the load and the `s_waitcnt vmcnt(N)` that retires it
(loads return in order, so a wait with N younger loads outstanding retires it).  The compiler is free to insert register copies,
so this script re-derives the property from the generated ISA of every k_fill instantiation and fails loudly if it is violated.
The same holds for the explicitly issued texel loads of k_raymarch's two-sample loop (vaddr form); there every 16-byte global
load of the kernel is tracked (the compiler's own loads satisfy the property by construction).
usage: check_fill_asm.py [fill|raymarch] [extra hipcc flags]      (compiles to asm with hipcc; exit code 1 on violation)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WHICH = sys.argv[1] if len(sys.argv) > 1 else "fill"
CSRC = os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc")
SRC = WHICH if WHICH.endswith(".hip") else os.path.join(CSRC, WHICH + ".hip")
KERNEL = "k_raymarch" if "raymarch" in os.path.basename(SRC) else "k_fill"
LOAD_RE = (r"global_load_dwordx4\s+(v\[\d+:\d+\]),\s*v\[\d+:\d+\],\s*off" if KERNEL == "k_raymarch"
           else r"global_load_dwordx4\s+(v\[\d+:\d+\]),\s*v\d+,\s*s\[\d+:\d+\]")
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
            mm = re.search(r"vmcnt\((\d+)\)", t)
            if mm:
                n = int(mm.group(1))
                outstanding = [] if n == 0 else outstanding[-n:]
            continue
        # younger compiler-issued VMEM ops only make our waits stricter (in-order return): safe to ignore
        touched = regs(t)
        for dest, lno in outstanding:
            if touched & dest:
                bad.append((no, t, lno))
    return bad


def main():
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
