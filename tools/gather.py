"""Gather floor vs window length and array size: python tools/gather.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
for recs, nwin in ((8, 1_000_000), (64, 1_000_000), (512, 1_000_000)):
    for arr in (1_000_000, 250_000, 62_500):
        r = C.c_double()
        _lib.check(_lib.lib().nhp_probe_gather(ctx.h, nwin, recs, arr, 2048, C.byref(r)), ctx.h)
        gb = nwin * recs * 16 / 1e9
        print(f"windows of {recs:4d} records out of {arr*16/1e6:6.1f} MB: {r.value:9.2f} us  ({gb / (r.value * 1e-6) / 1e3:6.2f} TB/s delivered)")
