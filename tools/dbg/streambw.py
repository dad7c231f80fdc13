"""Streaming-read ceiling (nhp_probe_stream): 16 bytes per lane and the child slices' two planes, by buffer size and grid."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
for mode in (0, 1):
    for mb in (48, 64, 128, 512):
        for blocks, threads in ((1024, 512), (2048, 256), (1024, 256), (512, 512)):
            r = C.c_double()
            _lib.check(_lib.lib().nhp_probe_stream(ctx.h, mode, mb * 1_000_000, blocks, threads, C.byref(r)), ctx.h)
            print(f"mode {mode} {mb:4d} MB  grid {blocks:4d} x {threads:3d}: {r.value:7.2f} us  {mb / r.value * 1e-3 * 1e3:6.2f} TB/s", flush=True)
