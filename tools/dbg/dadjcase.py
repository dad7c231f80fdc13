#!/usr/bin/env python3
"""One case of tests/test_discrete_gibbs_gpu.py::test_adjacency_sweep_over_long_spans outside pytest (a checker's tool: it
loads the oracle):  python tools/dbg/dadjcase.py N T B L rate spans big"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
nhp = entry.load_package()
from oracle import oracle as orc
orc.lib()
from test_discrete_gibbs_gpu import make_network
N, T, B, L = (int(v) for v in sys.argv[1:5])
rate, spans, big = float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
if spans:
    os.environ["NHP_DADJ_SPANS"] = str(spans)
import time
t0 = time.time()
proc, data = make_network(nhp, N, T, B, L, rate, seed=5 * N + T)
print('make_network', time.time() - t0)
if big:
    data = data.copy()
    data[N // 2, T // 3] = big
    if big >= 256:
        lam0 = proc.baseline.λ.copy(); lam0[N // 2] = 280.0; proc.baseline.λ = lam0
proc.weights.W = proc.weights.W * N * 1.5
t0 = time.time()
ds, conv = nhp.convolve(proc, data, fetch=True)
print('convolve', time.time() - t0)
u = np.random.default_rng(11).uniform(size=(N, N))
A0 = proc.adjacency_matrix.copy()
t0 = time.time()
want = orc.disc_resample_adjacency(data, conv, proc.baseline.λ, proc.weights.W, proc.impulses.θ, A0, 0.3, u, proc.dt)
print('oracle', time.time() - t0); t0 = time.time()
nhp.disc_resample_adjacency_matrix_(proc, convolved=ds, u=u)
print('gpu', time.time() - t0)
print("equal:", np.array_equal(proc.adjacency_matrix, want), " differing entries (p, c):", np.argwhere(proc.adjacency_matrix != want).tolist())
