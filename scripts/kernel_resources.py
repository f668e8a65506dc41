#!/usr/bin/env python3
"""Per-kernel register / scratch usage of a csrc/*.hip file as hipcc reports it (gfx950, cross-compiled; no GPU needed).
usage: kernel_resources.py fill.hip [extra hipcc flags...]   (e.g. -DVPFX_FILL_WAVES=3)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc", sys.argv[1])
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only",
                      "-S", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True)
if out.returncode:
    sys.exit(out.stderr[-3000:])
for b in re.split(r"Function Name: ", out.stderr)[1:]:
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{name:44s} VGPR {g('VGPRs'):3d}  SGPR {g('TotalSGPRs'):3d}  scratch {scratch:3d}  waves/SIMD {occ}"
          f"  SGPR-spill {g('SGPRs Spill'):3d}  VGPR-spill {g('VGPRs Spill'):3d}")
