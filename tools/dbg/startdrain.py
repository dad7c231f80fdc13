#!/usr/bin/env python3
"""What the bench's timed bracket costs besides its kernels: wall of (timer_start, K enqueues, timer_stop, torch sync) for K = 0, 1, 20, 200
at the metric size, with HIP's default wait policy and with hipDeviceScheduleSpin (SPIN=1, set before anything touches the device)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("SPIN"):
    hip = C.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin):", hip.hipSetDeviceFlags(C.c_uint(1)))
import numpy as np, torch
import __graft_entry__ as entry
nhp = entry.load_package()
from nhp_amd import _lib
lib = _lib.lib()
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
ds = nhp.continuous.DeviceDataset(ctx, (times, nodes, T), N, 1.0)
model = proc.device_model(ctx)
def enqueue(k):
    _lib.check(lib.nhp_cont_loglik_enqueue(ctx.h, ds.h, model.h, 0, k % _lib.MAX_SLOTS), ctx.h)
for k in range(10): enqueue(k)
ctx.synchronize(); torch.cuda.synchronize()
for K in (0, 1, 20, 200):
    best = []
    for rep in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.timer_start()
        for k in range(K): enqueue(k)
        dev = ctx.timer_stop()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best.append((t2 - t0, t1 - t0, dev))
    best.sort()
    w, w1, dev = best[len(best) // 2]
    print(f"K={K:4d}: wall {1e6*w:8.1f} us (to timer_stop {1e6*w1:8.1f}), hipEvents {1e3*dev:8.1f} us, wall - events {1e6*w - 1e3*dev:6.1f} us", flush=True)
