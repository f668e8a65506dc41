import sys, os, subprocess
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 2:
    import numpy as np
    from __graft_entry__ import load_package
    load_package()
    from vpfx_amd import abi, engine as E, scene as S
    N, nv, P, exact = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sc = S.make_scene("x", dims=(N, nv, P, 64, 64))
    g = E.Engine(sc.config(), exact=bool(exact))
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g.fill(sc.fill_params()); g.sync()
    print("ok", g.stats()["occupied_mv"], g.stats()["max_pairs_per_mv"], round(g.last_kernel_ms(1), 3), flush=True)
else:
    for args in ["32 32 100000 0", "32 32 100000 1", "32 32 10000 0", "24 32 40000 0", "20 32 25000 0", "32 16 100000 0", "16 32 100000 0"]:
        r = subprocess.run([sys.executable, __file__] + args.split(), capture_output=True, text=True)
        out = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l]
        print(args, "-> rc", r.returncode, out[-1][:160] if out else "")
