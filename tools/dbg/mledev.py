#!/usr/bin/env python3
"""The device-resident mle! alone at the metric size (windowed objective): ms per step.  Usage: tools/dbg/mledev.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
nhp = entry.load_package()
N, M = 1024, 1_000_000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = nhp.Context(0)
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
for rep in range(2):
    proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
    guess = np.clip(proc.params() * np.random.default_rng(9).uniform(0.5, 1.5, len(proc.params())), 1e-6, 10.0)
    nhp.device_dataset(proc, (times, nodes, T), ctx)
    t0 = time.perf_counter()
    res = nhp.mle_(proc, (times, nodes, T), guess=guess, recursive=bool(int(os.environ.get("REC", "0"))), f_abstol=1e-12, max_steps=steps, optimizer="device", ctx=ctx)
    dt = time.perf_counter() - t0
    print(f"device mle!: {res.steps} steps, {res.evaluations} evaluations in {dt:.3f} s = {1e3 * dt / max(1, res.steps):.3f} ms per step, "
          f"log-likelihood {res.maximum:.6f} ({res.status})", flush=True)
