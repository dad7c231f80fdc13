"""k_disc_convolve at BASELINE config 4 (N=512, B=8, L=32, T=1e5): hipEvent time of nhp_disc_convolve's kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
ctx = nhp.Context(0)
N, B, L, T = 512, 8, 32, 100_000
rng = np.random.default_rng(7)
data = rng.poisson(float(os.environ.get("RATE", 0.05)), (N, T)).astype(np.int64)
th = np.asfortranarray(np.full((N, N, B), 1.0 / B))
proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(rng.uniform(0.02, 0.08, N), 1.0), nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                         nhp.DenseWeightModel(np.asfortranarray(rng.uniform(0, 1, (N, N)) / N)), 1.0)
ds = nhp.DiscreteDataset(ctx, data)
nhp.convolve(proc, ds, ctx=ctx)
ts = []
for _ in range(5):
    ctx.synchronize(); t0 = time.perf_counter(); nhp.convolve(proc, ds, ctx=ctx); ctx.synchronize(); ts.append(time.perf_counter() - t0)
print(f"convolve call (k_disc_convolve + k_disc_convsum + basis upload): {1e3*min(ts):.3f} ms; 3.28 GB written -> {3.2768e9/min(ts)/1e12:.2f} TB/s on the whole call")
