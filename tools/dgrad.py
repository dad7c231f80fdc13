"""Discrete ll + gradient (the mle! objective) at config-4 scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
ctx = nhp.Context(0)
N, B, L, T = 512, 8, 32, 100_000
rng = np.random.default_rng(7)
data = np.asfortranarray(rng.poisson(0.05, (N, T)).astype(np.int64))
th = np.asfortranarray(np.full((N, N, B), 1.0 / B))
proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(rng.uniform(0.02, 0.08, N), 1.0),
                                         nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                         nhp.DenseWeightModel(np.asfortranarray(rng.uniform(0, 1, (N, N)) / N)), 1.0)
ds = nhp.convolve(proc, data, ctx=ctx)
for s in range(4):
    t0 = time.perf_counter()
    ll, g = nhp.loglikelihood_gradient(proc, data, convolved=ds, ctx=ctx)
    print(f"ll + gradient {1e3*(time.perf_counter()-t0):.1f} ms   ll={ll:.6f} |g|={np.linalg.norm(g):.6e}", flush=True)
