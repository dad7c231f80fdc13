#!/usr/bin/env python3
"""One-off set-up times: discrete dataset creation + convolution at config-4 scale, continuous dataset + first evaluation at the metric size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
ctx = nhp.Context(0)
N, B, L, T = 512, 8, 32, 100_000
rng = np.random.default_rng(7)
data = rng.poisson(0.05, (N, T)).astype(np.int64)
dataF = np.asfortranarray(data)
for name, d in (("C-order input", data), ("Fortran-order input", dataF)):
    t0 = time.perf_counter(); ds = nhp.DiscreteDataset(ctx, d); ctx.synchronize(); t1 = time.perf_counter()
    print(f"DiscreteDataset ({name}, N={N}, T={T}): {1e3*(t1-t0):.1f} ms", flush=True)
times, nodes, Tc = nhp.synthetic.s_metric_data(1024, 1_000_000, kbar=8.0)
proc = nhp.synthetic.s_metric_process(1024, 1_000_000, Tc, "exponential", 1.0)
t0 = time.perf_counter(); ll = nhp.loglikelihood(proc, (times, nodes, Tc), recursive=False, ctx=ctx); t1 = time.perf_counter()
print(f"first loglikelihood of fresh data (upload, layout, model upload, evaluation): {1e3*(t1-t0):.1f} ms  ll={ll:.3f}", flush=True)
t0 = time.perf_counter(); ll = nhp.loglikelihood(proc, (times, nodes, Tc), recursive=False, ctx=ctx); t1 = time.perf_counter()
print(f"second call (cached dataset, model re-uploaded): {1e3*(t1-t0):.2f} ms", flush=True)
