"""Discrete Gibbs parent-count sweep at BASELINE config 4 (N=512, B=8, L=32, T=1e5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
ctx = nhp.Context(0)
N, B, L, T = 512, 8, 32, 100_000
rate = float(os.environ.get("DG_RATE", 0.1))
rng = np.random.default_rng(0)
data = np.asfortranarray(rng.poisson(rate, (N, T)).astype(np.int64))
th = np.full((N, N, B), 1.0 / B)
proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(np.full(N, 0.05), 1.0),
                                         nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                         nhp.DenseWeightModel(np.full((N, N), 0.5 / N)), 1.0)
ds = nhp.convolve(proc, data, ctx=ctx)
for s in range(3):
    t0 = time.perf_counter()
    c = nhp.resample_parent_counts(proc, convolved=ds, seed=1, step=s, ctx=ctx)
    print(f"parent-count sweep {1e3*(time.perf_counter()-t0):.1f} ms  events={int(data.sum())} placed={int(c.sum())} baseline share={c[:,0].sum()/c.sum():.3f}", flush=True)
