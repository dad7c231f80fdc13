"""Parent-sampler timing over window lengths: python tools/samp.py  (NHP_SAMPLER_COOP=0|1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
for kind in ("exponential", "logitnormal"):
    for kbar in (8.0, 64.0, 512.0):
        times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=kbar)
        proc = nhp.synthetic.s_metric_process(N, M, T, kind, 1.0, network=True)
        ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
        best = 1e9
        for s in range(6):
            t0 = time.perf_counter()
            nhp.resample_parents(proc, ds, seed=1, step=s, with_stats=True, want_parents=False, ctx=ctx)
            best = min(best, time.perf_counter() - t0)
        print(f"{kind:12s} K={kbar:5.0f}  sampler+stats {1e3*best:8.3f} ms", flush=True)
        del ds
