"""CPU: the fp32 C oracle against the independent float64 numpy twin on the random scenes of fuzz_parity.py (small ones: the twin is
slow).  The two differ legitimately where a lattice sample or a voxel centre sits exactly on a decision boundary (fp32 vs fp64), so
the check is statistical: identical bin counts, light map to 1e-4, sample counts within 1e-4 relative, and all but a few pixels
within 1e-4.   usage: fuzz_twin.py [cases] [first_seed]"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
from oracle import oracle as O  # noqa: E402
from oracle.numpy_twin import Twin  # noqa: E402


def one(seed):
    sc, _ = fz.make_case_scene(seed)
    if sc.nv != 16 or max(sc.N) > 4 or len(sc.particles) > 400:
        return None
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    img = o.raymarch(sc.camera(), sc.raymarch_params())
    tw = Twin(sc)
    tw.grid(); tw.bin(); tw.fill()
    timg = tw.raymarch()
    d = np.abs(timg - img).max(axis=-1)
    so = o.stats()["samples"]
    return dict(seed=seed, N=sc.N, P=len(sc.particles), border=sc.border, samples=(int(tw.samples), int(so)),
                max=float(d.max()), n_gt_1e4=int((d > 1e-4).sum()), npix=int(d.size))


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    done, seed, t0 = 0, first, time.time()
    while done < cases:
        r = one(seed)
        seed += 1
        if r is None:
            continue
        done += 1
        ok = abs(r["samples"][0] - r["samples"][1]) <= 1e-4 * r["samples"][1] + 4 and r["n_gt_1e4"] <= max(3, r["npix"] // 2000)
        print(("ok  " if ok else "DIFF"), r, flush=True)
    print(f"{cases} cases in {time.time() - t0:.0f} s")
