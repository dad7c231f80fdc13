#!/usr/bin/env python3
"""Is the slice gradient the same bits run to run / build to build?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
nhp = entry.load_package()
from helpers import random_case
os.environ["NHP_CHUNK"] = "4096"
N, M = 1040, 60000
c = random_case(N, M, 4000.0, "exponential", 1.0, seed=31, nhp=nhp)
data = c["data"]
import ctypes as C, hashlib
from nhp_amd import _lib
def planes():
    ds = nhp.device_dataset(c["proc"], data)
    fn = _lib.lib().nhp_debug_parent_slices
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    n = fn(ds.h, None, None, 0)
    lo = np.zeros(n, np.uint32); hi = np.zeros(n, np.uint16)
    assert fn(ds.h, lo.ctypes.data, hi.ctypes.data, n) == n
    return hashlib.md5(lo.tobytes() + hi.tobytes()).hexdigest()[:12], n
ll0, g0 = nhp.loglikelihood_gradient(c["proc"], data, recursive=False)
print("planes", planes())
for trial in range(3):
    ll, g = nhp.loglikelihood_gradient(c["proc"], data, recursive=False)
    d = np.flatnonzero(g != g0)
    print("same dataset:", len(d), "entries differ", d[:5], "planes", planes())
for trial in range(3):
    nhp.invalidate_device_datasets()
    ll, g = nhp.loglikelihood_gradient(c["proc"], data, recursive=False)
    d = np.flatnonzero(g != g0)
    blk = np.where(d < N, 0, np.where(d < N + N * N, 1, 2))
    print("planes", planes())
    print("rebuilt:", len(d), "entries differ; by block (λ0, θ, W):", np.bincount(blk, minlength=3), d[:5], (g - g0)[d[:5]], g0[d[:5]])
