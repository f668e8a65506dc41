import ctypes as C, time, sys
sys.path.insert(0,'.')
from __graft_entry__ import load_package; load_package()
from vpfx_amd import engine as E
E.lib()
HIP = C.CDLL("libamdhip64.so.7")
s = C.c_void_p()
print("create", HIP.hipStreamCreateWithFlags(C.byref(s), 1))
CB = C.CFUNCTYPE(None, C.c_void_p)
def cb(p):
    time.sleep(0.5)
cbf = CB(cb)
t0=time.perf_counter()
print("launch rc", HIP.hipLaunchHostFunc(s, cbf, None))
n=0
while HIP.hipStreamQuery(s) != 0:
    n+=1
print("polls", n, "elapsed", time.perf_counter()-t0)
