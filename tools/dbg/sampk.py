#!/usr/bin/env python3
"""The device-resident Gibbs sweep (parents, statistics, draws) at the metric size for both impulse families: under rocprofv3
(tools/kstats.sh) the sampler kernels' times -- exponential k_sampler<0> against logit-normal k_sampler_slices."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib, inference
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=float(os.environ.get("KBAR", 8.0)))
for kind in os.environ.get("KINDS", "exponential,logitnormal").split(","):
    proc = nhp.synthetic.s_metric_process(N, M, T, kind, 1.0, network=True)
    ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
    model, pri = proc.device_model(ctx), inference._priors(proc)
    for s in range(20):
        _lib.check(_lib.lib().nhp_cont_gibbs_step(ctx.h, ds.h, model.h, C.byref(pri), 1, s), ctx.h)
    ctx.synchronize()
    print(kind, "done", flush=True)
