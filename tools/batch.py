"""Batched log-likelihood (nhp_cont_loglik_batch): NB parameter sets on one dataset.  NHP_BATCH_KERNEL=0 selects the older
k_windowed_multi, NHP_BATCH_FUSE the largest group, NHP_BATCH_LANES=1 one stream; KBAR the mean window; KIND the impulse."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
N, M = int(os.environ.get("KB_N", 1024)), int(os.environ.get("KB_M", 1_000_000))
kind = os.environ.get("KIND", "exponential")
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=float(os.environ.get("KBAR", 8.0)))
procs = []
for s in range(8):
    p = nhp.synthetic.s_metric_process(N, M, T, kind, 1.0)
    p.weights.W = p.weights.W * (1.0 + 0.01 * s)
    procs.append(p)
ds = nhp.device_dataset(procs[0], (times, nodes, T), ctx)
models = [p.device_model(ctx) for p in procs]
single = [nhp.loglikelihood(p, ds, recursive=False, ctx=ctx) for p in procs]
for NB in [int(v) for v in os.environ.get('NB', '8,32,64').split(',')]:
    arr = (C.c_void_p * NB)(*[models[i % 8].h for i in range(NB)])
    out = np.empty(NB)
    best = 1e9
    for rep in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, NB, 0, _lib.dptr(out)), ctx.h)
        best = min(best, (time.perf_counter() - t0) / 20)
    err = max(abs(out[i] - single[i % 8]) / abs(single[i % 8]) for i in range(NB))
    print(f"batch of {NB:3d}: {1e6*best:8.1f} us  = {1e6*best/NB:6.2f} us per evaluation  ({NB/best:9.0f} evals/s)  max rel diff vs single {err:.1e}", flush=True)
