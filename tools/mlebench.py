#!/usr/bin/env python3
"""mle! end to end at a given size: the host optimizer (scipy L-BFGS-B on the GPU's analytic gradient: x up, gradient down and a
host-side update per objective call) against the device-resident one (nhp_cont_mle_run).  Usage: tools/mlebench.py [N M steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
nhp = entry.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ctx = nhp.Context(0)
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
for recursive in (False, True):
    out = {}
    for opt in ("L-BFGS-B", "device"):
        proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
        guess = np.clip(proc.params() * np.random.default_rng(9).uniform(0.5, 1.5, len(proc.params())), 1e-6, 10.0)
        nhp.device_dataset(proc, (times, nodes, T), ctx)
        t0 = time.perf_counter()
        res = nhp.mle_(proc, (times, nodes, T), guess=guess, recursive=recursive, f_abstol=1e-12, max_steps=steps, optimizer=opt, ctx=ctx)
        dt = time.perf_counter() - t0
        out[opt] = (dt, res)
        ev = getattr(res, "evaluations", None)
        print(f"N={N} M={M} recursive={recursive!s:5s} {opt:9s}: {res.steps:4d} steps in {dt:8.3f} s = {1e3 * dt / max(1, res.steps):8.2f} ms per step"
              f"{'' if ev is None else f' ({ev} evaluations)'}, log-likelihood {res.maximum:.6f} ({res.status})", flush=True)
