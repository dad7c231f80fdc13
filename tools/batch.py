"""Batched log-likelihood (nhp_cont_loglik_batch): S parameter sets on one dataset, fused vs one launch each."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
procs = []
for s in range(8):
    p = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
    p.weights.W = p.weights.W * (1.0 + 0.01 * s)
    procs.append(p)
ds = nhp.device_dataset(procs[0], (times, nodes, T), ctx)
models = [p.device_model(ctx) for p in procs]
NB = int(os.environ.get('NB', 8))
arr = (C.c_void_p * NB)(*[models[i % 8].h for i in range(NB)])
out = np.empty(NB)
for rep in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, NB, 0, _lib.dptr(out)), ctx.h)
    dt = (time.perf_counter() - t0) / 20
    print(f"batch of {NB}: {1e6*dt:8.1f} us  = {1e6*dt/NB:6.1f} us per evaluation   ll[0]={out[0]:.6f} ll[7]={out[7]:.6f}", flush=True)
