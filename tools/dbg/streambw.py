"""Streaming-read ceiling (nhp_probe_stream): 16 bytes per lane and the child slices' two planes, by buffer size and grid.
STREAM_ONE="mode,MB,blocks,threads": that one configuration only (the FETCH_SIZE calibration pass of tools/traffic.sh)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
one = os.environ.get("STREAM_ONE")
cases = [tuple(int(v) for v in one.split(","))] if one else [(mode, mb, b, t) for mode in (0, 1) for mb in (48, 64, 128, 512)
                                                              for b, t in ((1024, 512), (2048, 256), (1024, 256), (512, 512))]
for mode, mb, blocks, threads in cases:
    r, n = C.c_double(), C.c_int64()
    _lib.check(_lib.lib().nhp_probe_stream(ctx.h, mode, mb * 1_000_000, blocks, threads, C.byref(r), C.byref(n)), ctx.h)
    print(f"mode {mode} {n.value / 1e6:7.1f} MB read  grid {blocks:4d} x {threads:3d}: {r.value:7.2f} us  {n.value / r.value * 1e-6:6.2f} TB/s  bytes={n.value}", flush=True)
