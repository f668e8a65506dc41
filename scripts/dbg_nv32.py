import sys, os, subprocess
sys.path.insert(0, "/root/repo")
import numpy as np
if len(sys.argv) > 1:
    from __graft_entry__ import load_package
    load_package()
    from vpfx_amd import abi, engine as E, scene as S
    from oracle import oracle as O
    case = sys.argv[1]
    dims = {"a": (8, 32, 1000, 64, 64), "b": (4, 16, 700, 64, 48), "c": (16, 32, 10000, 64, 64), "d": (4, 32, 700, 64, 48), "e": (8, 32, 6000, 64, 64)}[case]
    sc = S.make_scene("x", dims=dims)
    if case in "bd": sc.particles["position"] *= 0.15
    o, g = O.Oracle(sc.config()), E.Engine(sc.config(), exact=False, early_out=False)
    for x in (o, g):
        x.set_frame(sc.light_to_world, sc.grid_center); x.bin(sc.particles, sc.layout, sc.psys_local_to_world); x.fill(sc.fill_params())
    g.sync()
    cnt = o.bin_counts(); worst = 0
    for zz, yy, xx in list(zip(*np.nonzero(cnt)))[::3]:
        a, b = o.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32), g.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32)
        worst = max(worst, int(np.abs(a - b).max()))
    print(case, dims, "max pairs", g.stats()["max_pairs_per_mv"], "worst brick ulp diff", worst, flush=True)
else:
    for c in "abcde":
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
        print(c, "rc", r.returncode, (r.stdout + r.stderr).strip().splitlines()[-1][:200])
