#!/usr/bin/env python3
"""Do two independent evaluation streams on ONE GPU (two nhp_ctx = two HIP streams, e.g. two chains) overlap?
   python tools/twostream.py [n_streams] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
nhp = entry.load_package()
from nhp_amd import _lib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=float(os.environ.get("KB_K", 8)))
lib = _lib.lib()
ctxs, dss, models = [], [], []
for s in range(S):
    ctx = nhp.Context(0)
    proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
    ctxs.append(ctx); dss.append(nhp.continuous.DeviceDataset(ctx, (times, nodes, T), N, 1.0)); models.append(proc.device_model(ctx))
    setattr(proc, "_keep", True); models[-1]._proc = proc
def run(active, steps):
    for k in range(5):
        for s in active: _lib.check(lib.nhp_cont_loglik_enqueue(ctxs[s].h, dss[s].h, models[s].h, 0, k % _lib.MAX_SLOTS), ctxs[s].h)
    for s in active: ctxs[s].synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        for s in active: _lib.check(lib.nhp_cont_loglik_enqueue(ctxs[s].h, dss[s].h, models[s].h, 0, k % _lib.MAX_SLOTS), ctxs[s].h)
    for s in active: ctxs[s].synchronize()
    return (time.perf_counter() - t0)
t1 = run([0], steps)
print(f"1 stream : {steps / t1:10.0f} evals/s  ({1e6 * t1 / steps:.1f} us per evaluation)")
tS = run(list(range(S)), steps)
print(f"{S} streams: {S * steps / tS:10.0f} evals/s  ({1e6 * tS / (S * steps):.1f} us per evaluation)  ll={ctxs[0].fetch(0, 1)[0]:.6f} {ctxs[-1].fetch(0, 1)[0]:.6f}")
