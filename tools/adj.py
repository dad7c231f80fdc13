import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import inference
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
proc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
model = proc.device_model(ctx)
for s in range(4):
    t0 = time.perf_counter()
    inference.resample_adjacency_matrix_(proc, ds, seed=1, step=s, model=model, fetch=False, ctx=ctx)
    print(f"adjacency wall {1e3*(time.perf_counter()-t0):.2f} ms")
