#!/usr/bin/env python3
"""Discrete mle! end to end at config-4 scale (N=512, B=8, L=32, T=1e5 by default): host optimizer (scipy L-BFGS-B on the GPU's
analytic gradient) against the device-resident one (nhp_disc_mle_run).  Usage: tools/dmlebench.py [N T steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
nhp = entry.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, L = 8, 32
ctx = nhp.Context(0)
rng = np.random.default_rng(7)
data = rng.poisson(0.05, (N, T)).astype(np.int64)
guess = np.concatenate([rng.uniform(0.02, 0.08, N), rng.uniform(0.0, 1.0, N * N * B) / (N * B)])
for opt, n in (("device", steps), ("L-BFGS-B", min(steps, 3))):
    th = np.asfortranarray(np.full((N, N, B), 1.0 / B))
    proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(rng.uniform(0.02, 0.08, N), 1.0), nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                             nhp.DenseWeightModel(np.asfortranarray(rng.uniform(0, 1, (N, N)) / N)), 1.0)
    ds = nhp.convolve(proc, data, ctx=ctx)
    t0 = time.perf_counter()
    res = nhp.mle_(proc, ds, guess=guess, f_abstol=1e-12, max_steps=n, optimizer=opt, ctx=ctx)
    dt = time.perf_counter() - t0
    ev = getattr(res, "evaluations", None)
    print(f"N={N} T={T} {opt:9s}: {res.steps:4d} steps in {dt:8.3f} s = {1e3 * dt / max(1, res.steps):9.2f} ms per step"
          f"{'' if ev is None else f' ({ev} evaluations)'}, log-likelihood {res.maximum:.4f} ({res.status})", flush=True)
