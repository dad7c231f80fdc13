import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
ctx = nhp.Context(0)
N, B, L, T = 512, 8, 32, int(os.environ.get("C4_T", 100_000))
rng = np.random.default_rng(7)
data = rng.poisson(0.05, (N, T)).astype(np.int64)
imp = nhp.DiscreteGaussianImpulseResponse(np.full((N, N, B), 1.0 / B), L, 1.0)
proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(rng.uniform(0.02, 0.08, N), 1.0), imp,
                                         nhp.DenseWeightModel(rng.uniform(0, 1, (N, N)) / N), 1.0)
ds = nhp.DiscreteDataset(ctx, data)
nhp.convolve(proc, ds, ctx=ctx)
for _ in range(3):
    ll = nhp.loglikelihood(proc, data, convolved=ds, ctx=ctx)
for _ in range(3):
    nhp.update_(proc, data, ds, ctx=ctx)
print("ll", ll)
