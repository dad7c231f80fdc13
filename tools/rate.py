import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
for mode, name in ((0, "exp pair terms/s"), (1, "fp64 fma/s")):
    for blocks in (1024, 2048, 8192):
        r = C.c_double()
        _lib.check(_lib.lib().nhp_probe_rate(ctx.h, mode, 40000, blocks, C.byref(r)), ctx.h)
        print(f"{name:18s} blocks={blocks:5d}  {r.value:.4e}")

# gather floor of the short-window regime: 1e6 windows of 8 records (128 B) scattered over 1e6 records (16 MB)
for recs in (8, 64):
    for blocks in (1024, 2048, 4096):
        r = C.c_double()
        _lib.check(_lib.lib().nhp_probe_gather(ctx.h, 1_000_000, recs, 1_000_000, blocks, C.byref(r)), ctx.h)
        print(f"gather 1e6 windows x {recs:3d} records  blocks={blocks:5d}  {r.value:8.2f} us")
