"""Discrete adjacency Gibbs sweep at BASELINE config 4 scale (N=512, B=8, L=32, T=1e5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
ctx = nhp.Context(0)
N, B, L, T = 512, 8, 32, 100_000
rng = np.random.default_rng(0)
data = np.asfortranarray(rng.poisson(float(os.environ.get("DG_RATE", 0.1)), (N, T)).astype(np.int64))
th = np.asfortranarray(np.full((N, N, B), 1.0 / B))
A = (rng.uniform(size=(N, N)) < float(os.environ.get("A_DENSITY", 0.5))).astype(np.float64)
proc = nhp.DiscreteNetworkHawkesProcess(nhp.DiscreteHomogeneousProcess(np.full(N, 0.05), 1.0),
                                        nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                        nhp.DenseWeightModel(np.full((N, N), 1.0 / N)), A,
                                        nhp.BernoulliNetworkModel(0.5, N), 1.0)
ds = nhp.convolve(proc, data, ctx=ctx)
for s in range(3):
    t0 = time.perf_counter()
    links = nhp.disc_resample_adjacency_matrix_(proc, convolved=ds, seed=1, step=s, ctx=ctx)
    print(f"adjacency sweep {1e3*(time.perf_counter()-t0):.1f} ms  links={links:.0f} of {N*N}", flush=True)
