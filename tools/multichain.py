#!/usr/bin/env python3
"""Several mcmc! chains of the config-3 model on ONE GPU, one host thread + one nhp_ctx each: the sweeps of a single
chain leave most of the chip idle (the adjacency sweep runs one wave per column), so chains overlap.
   python tools/multichain.py [n_chains] [steps]"""
import os, sys, time, threading, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
nhp = entry.load_package()
from nhp_amd import _lib, inference, chains

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
work = []
for k in range(K):
    ctx = nhp.Context(0)
    proc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
    ds = nhp.continuous.DeviceDataset(ctx, (times, nodes, T), N, 1.0)
    model, pri = proc.device_model(ctx), inference._priors(proc)
    work.append((ctx, proc, ds, model, pri, chains.chain_seed(1, k)))

def run(w, n, first):
    ctx, proc, ds, model, pri, seed = w
    for s in range(first, first + n):
        _lib.check(_lib.lib().nhp_cont_gibbs_step(ctx.h, ds.h, model.h, C.byref(pri), seed, s), ctx.h)
        inference.resample_adjacency_matrix_(proc, ds, seed=seed, step=s, model=model, fetch=False, ctx=ctx)
        _lib.check(_lib.lib().nhp_cont_model_moments_accumulate(ctx.h, model.h), ctx.h)
    ctx.synchronize()

for w in work:
    run(w, 3, 0)
for active in ([work[0]], work):
    ths = [threading.Thread(target=run, args=(w, steps, 3)) for w in active]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print(f"{len(active)} chain(s): {len(active) * steps / dt:8.0f} mcmc steps/s in total ({1e3 * dt / steps:.3f} ms per step and chain)", flush=True)
