import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
kbar = float(os.environ.get("C3_K", 8))
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=kbar)
proc = nhp.synthetic.s_metric_process(N, M, T, os.environ.get("C3_KIND", "logitnormal"), 1.0, network=True)
ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
for s in range(3):
    t0 = time.perf_counter()
    nhp.resample_parents(proc, ds, seed=1, step=s, with_stats=True, want_parents=False, ctx=ctx)
    t1 = time.perf_counter()
    nhp.resample_parents(proc, ds, seed=1, step=s, with_stats=False, want_parents=True, ctx=ctx)
    t2 = time.perf_counter()
    print(f"stats-only {1e3*(t1-t0):.2f} ms   parents-only {1e3*(t2-t1):.2f} ms")
