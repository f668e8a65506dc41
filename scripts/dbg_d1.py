import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from __graft_entry__ import load_package; load_package()
from vpfx_amd import engine as E, scene as S
from oracle import oracle as O
rng=np.random.default_rng(3)
nv=int(sys.argv[1]) if len(sys.argv)>1 else 16
sc=S.make_scene("d", dims=(2,nv,300,32,32), border=1)
cube=rng.integers(0,256,size=(6,128,128),dtype=np.uint8)
sc.cubemap=cube; sc.displacement_scale=1.0
o=O.Oracle(sc.config())
def run(cfg):
    g=E.Engine(cfg)
    for x in (g,):
        x.set_frame(sc.light_to_world, sc.grid_center); x.bin(sc.particles, sc.layout, sc.psys_local_to_world); x.fill(sc.fill_params())
    return g
o.set_frame(sc.light_to_world, sc.grid_center); o.bin(sc.particles, sc.layout, sc.psys_local_to_world); o.fill(sc.fill_params())
cfg2=sc.config(); cfg2.reserved[0]=1
for label,cfg in (("lds",sc.config()),("global",cfg2)):
    g=run(cfg)
    cnt=o.bin_counts(); tot=0; bad=0; worst=None
    for zz,yy,xx in zip(*np.nonzero(cnt)):
        fa,fb=o.read_brick(xx,yy,zz).astype(np.float32),g.read_brick(xx,yy,zz).astype(np.float32)
        d=np.abs(fa-fb); tot+=d.size//4
        b=(d[...,3]>1e-3); bad+=int(b.sum())
        if b.any() and worst is None:
            idx=np.unravel_index(np.argmax(d[...,3]),d[...,3].shape); worst=((xx,yy,zz),idx,fa[idx],fb[idx])
            s,py,px=idx
            print(label,"column dens oracle",fa[:,py,px,3]); print(label,"column dens gpu   ",fb[:,py,px,3])
    print(label,"voxels",tot,"bad density voxels",bad,"first worst",worst, "nan", flush=True)
    g.close()
