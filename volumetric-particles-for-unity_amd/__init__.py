"""vpfx_amd -- MI355X-native sparse volumetric particle fill + ray-march (hot path only).

The directory is named after the reference repo (`volumetric-particles-for-unity_amd`), which is not a
valid Python identifier; `__graft_entry__.load_package()` imports it under the module name `vpfx_amd`.

Contents: `csrc/` (hand-written HIP kernels for gfx950 + the C ABI of include/vpfx.h), `abi.py` (ctypes
mirror of the header), `engine.py` (thin ctypes binding; fails loudly when libvpfx is missing),
`manager.py` (host-side mirror of the reference's VolumetricParticleRenderer interface),
`scene.py` (synthetic workload),  (the multi-GPU fan-out lives inside the library: csrc/multi.cpp),
`csharp/` (the C# P/Invoke shim a Unity maintainer would add; source only, no C# toolchain here).
"""
__all__ = ["abi", "scene"]
