"""Probe (round 6): two contexts on two HIP streams, frames alternating between them -- does the GPU overlap one frame's ray-march (latency-bound, VALU
0.57) with the next frame's fill (VALU-issue-bound)?  Every frame is still a full bin + fill + ray-march of its own context; only the stream differs.
usage: pipeline2_probe.py [C3] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from vpfx_amd import engine as E, scene as S

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sc = S.make_scene(name, cubemap="r8")
dev = torch.device("cuda", 0)
cam, rp = sc.camera(), sc.raymarch_params()


def make(stream):
    e = E.Engine(sc.config())
    if stream is not None:
        e.set_stream(stream.cuda_stream)
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
    fp = sc.fill_params()
    e.bin_resident(); e.fill(fp)
    fp2 = sc.fill_params(); fp2.cubemap = None
    img = torch.empty((sc.height, sc.width, 4), device=dev)
    e.raymarch_device(cam, rp, img.data_ptr())
    e.sync()
    return e, fp2, img


def run(ctxs, n):
    for e, _, _ in ctxs:
        e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        e, fp, img = ctxs[i % len(ctxs)]
        e.bin_resident(); e.fill(fp); e.raymarch_device(cam, rp, img.data_ptr())
    for e, _, _ in ctxs:
        e.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = [make(None)]
run(one, 10)
t1 = run(one, steps)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
two = [make(s1), make(s2)]
run(two, 10)
t2 = run(two, steps)
ref = one[0][2]
err = max(float((c[2] - ref).abs().max().item()) for c in two)
print(f"{name}: one context {t1:.3f} ms/step; two contexts on two streams, frames alternating {t2:.3f} ms/step ({t1 / t2:.3f}x); images equal to {err:.1e}")
