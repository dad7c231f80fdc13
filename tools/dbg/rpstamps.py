#!/usr/bin/env python3
"""Walk 1 of the discrete parent kernel, where a workgroup's time goes (a -DRP_STAMP build, NHP_LIB=...): s_memtime cycles of
thread 0 summed over the chunks -- the barrier before a chunk is staged, the wait for its loads, staging + barrier, arithmetic."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
N, B, L, T = 512, 8, 32, 100_000
rng = np.random.default_rng(0)
data = np.asfortranarray(rng.poisson(0.05, (N, T)).astype(np.int64))
proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(np.full(N, 0.05), 1.0),
                                         nhp.DiscreteGaussianImpulseResponse(np.full((N, N, B), 1.0 / B), L, 1.0),
                                         nhp.DenseWeightModel(np.full((N, N), 0.5 / N)), 1.0)
ds = nhp.convolve(proc, data, ctx=ctx)
nhp.resample_parent_counts(proc, convolved=ds, seed=1, step=0, ctx=ctx)
out = (C.c_ulonglong * (8 * 3136))()
fn = C.CDLL(_lib.LIB_PATH).nhp_debug_rp_stamps
print("rc", fn(out, 8 * 3136))
a = np.array(out, dtype=np.float64).reshape(-1, 8)
a = a[a.sum(axis=1) > 0]
for w in (0, 1):
    b = a[:, 4 * w:4 * w + 4]
    tot = b.sum(axis=1)
    print(f"walk {w + 1}: workgroups {len(b)}; s_memtime ticks per workgroup, p50 of the total {np.median(tot):.0f}")
    for k, n in enumerate(["barrier before staging", "wait for the chunk's loads", "staging + barrier", "arithmetic"]):
        print(f"  {n:28s} p50 {np.median(b[:, k]):9.0f}  ({100 * np.median(b[:, k] / tot):4.1f} %)")
