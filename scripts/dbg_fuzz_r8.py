import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("fz", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec)
src = open(spec.origin).read().split("if __name__")[0]
exec(compile(src, spec.origin, "exec"), fz.__dict__)
from vpfx_amd import engine as E
from oracle import oracle as O
for seed in (20125, 20347):
    sc, rng = fz.make_case_scene(seed)
    o = O.Oracle(sc.config()); o.set_frame(sc.light_to_world, sc.grid_center); o.bin(sc.particles, sc.layout, sc.psys_local_to_world); o.fill(sc.fill_params())
    cnt = o.bin_counts()
    res = []
    for nolds in (0, 1):
        cfg = sc.config(); cfg.reserved[0] = nolds
        g = E.Engine(cfg); g.set_frame(sc.light_to_world, sc.grid_center); g.bin(sc.particles, sc.layout, sc.psys_local_to_world); g.fill(sc.fill_params())
        worst, where = 0, None
        for zz, yy, xx in zip(*np.nonzero(cnt)):
            a = o.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32); b = g.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32)
            dd = np.abs(a - b)
            if dd.max() > worst:
                worst = int(dd.max()); idx = np.unravel_index(np.argmax(dd), dd.shape); where = (xx, yy, zz, idx, o.read_brick(xx, yy, zz)[idx[:3]], g.read_brick(xx, yy, zz)[idx[:3]])
        res.append((worst, where))
    print(seed, "S", sc.cubemap.shape[1], sc.cubemap.dtype, "D", sc.displacement_scale, "nv", sc.nv, "LDS:", res[0][0], "global:", res[1][0])
    print("    ", res[0][1])
