#!/usr/bin/env python3
"""log-likelihood + analytic gradient at the metric size (the mle! objective), parameters resident."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
nhp = entry.load_package()
N, M = 1024, 1_000_000
ctx = nhp.Context(0)
for kbar in (8.0, 64.0):
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=kbar)
    for kind in ("exponential", "logitnormal"):
        proc = nhp.synthetic.s_metric_process(N, M, T, kind, 1.0)
        ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
        model = proc.device_model(ctx)
        for rec in ((False, True) if kind == "exponential" else (False,)):
            nhp.loglikelihood_gradient(proc, ds, recursive=rec, ctx=ctx, model=model)
            t0 = time.perf_counter()
            for _ in range(5):
                ll, g = nhp.loglikelihood_gradient(proc, ds, recursive=rec, ctx=ctx, model=model)
            dt = (time.perf_counter() - t0) / 5
            print(f"kbar={kbar:5.0f} {kind:12s} recursive={rec!s:5s}: {1e3*dt:8.2f} ms per (ll, gradient)  [P = {len(g)}]", flush=True)

# one mle! objective call as inference.mle_ issues it: params!(x) into the resident model, then (ll, gradient)
import numpy as np
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
model = proc.device_model(ctx)
x = proc.params()
for _ in range(2):
    model.set_params(x)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    model.set_params(x)
ctx.synchronize()
print(f"params!(x) upload of {8 * len(x) / 1e6:.1f} MB: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms", flush=True)
