#!/usr/bin/env python3
"""The sparse-truth mle! of tests/test_cont_inference_gpu.py::test_device_mle_with_most_weights_on_the_lower_bound, several runs:
steps, value reached and status of the device optimizer against the host optimizer's polish."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
nhp = entry.load_package()
N, T = 8, 600.0
rng = np.random.default_rng(12)
W = rng.uniform(0.1, 0.4, (N, N)) * (rng.uniform(size=(N, N)) < float(os.environ.get("LINKS", 0.25))) * float(os.environ.get("WSCALE", 1.0))
proc = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(rng.uniform(0.5, 1.0, N)),
                                           nhp.ExponentialImpulseResponse(rng.uniform(2.0, 4.0, (N, N)), 1.0, 1.0, 2.0),
                                           nhp.DenseWeightModel(W))
data = nhp.synthetic.rand(proc, T, seed=4)
guess = np.random.default_rng(6).uniform(0.3, 0.9, len(proc.params()))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    dev = nhp.mle_(proc, data, guess=guess, recursive=False, f_abstol=1e-10, max_steps=int(os.environ.get("MAXSTEPS", 5000)), optimizer="device")
    print(f"run {rep}: steps {dev.steps:5d} evaluations {dev.evaluations:6d} maximum {dev.maximum:.6f} {dev.status}", flush=True)
pol = nhp.mle_(proc, data, guess=dev.maximizer, recursive=False, f_abstol=1e-10, max_steps=3000)
print(f"host polish from the last: {pol.maximum:.6f} in {pol.steps} steps")
if os.environ.get("HOST"):
    import time
    t0 = time.time()
    host = nhp.mle_(proc, data, guess=guess, recursive=False, f_abstol=1e-10, max_steps=5000)
    print(f"host optimizer from the same start: {host.maximum:.6f} in {host.steps} steps ({host.status}), {time.time() - t0:.1f} s")
    # scipy's own tests off (its defaults stop at a RELATIVE decrease of 2.2e-9 or a projected gradient of 1e-5, long before
    # |f_k - f_{k-1}| < 1e-10 at |f| ~ 1e4): the same stopping rule as the device optimizer
    from scipy import optimize
    plain = optimize.minimize
    optimize.minimize = lambda *a, **k: plain(*a, **{**k, "options": {**k.get("options", {}), "ftol": 0.0, "gtol": 0.0, "maxcor": 8, "maxfun": 10**6}})
    t0 = time.time()
    host = nhp.mle_(proc, data, guess=guess, recursive=False, f_abstol=1e-10, max_steps=20000)
    print(f"host optimizer, its own tests off (m = 8): {host.maximum:.6f} in {host.steps} steps ({host.status}), {time.time() - t0:.1f} s")
