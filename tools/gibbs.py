import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import inference, _lib
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=float(os.environ.get("C3_K", 8)))
proc = nhp.synthetic.s_metric_process(N, M, T, os.environ.get("C3_KIND", "logitnormal"), 1.0, network=True)
ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
model, pri = proc.device_model(ctx), inference._priors(proc)
for s in range(6):
    t0 = time.perf_counter()
    _lib.check(_lib.lib().nhp_cont_gibbs_step(ctx.h, ds.h, model.h, C.byref(pri), 1, s), ctx.h)
    ctx.synchronize()
    print(f"gibbs step wall {1e3*(time.perf_counter()-t0):.3f} ms")
