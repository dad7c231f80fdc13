import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
nhp = entry.load_package()
from oracle import oracle as orc
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
proc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
om = orc.ContModel(proc.baseline.λ, proc.weights.W, mu=proc.impulses.μ, tau=proc.impulses.τ, dt_max=1.0, A=proc.adjacency_matrix)
A0 = proc.adjacency_matrix.copy()
proc.network.ρ = 0.35
u = np.random.default_rng(5).uniform(size=(N, N))
links = nhp.resample_adjacency_matrix_(proc, (times, nodes, T), u=u)
got = proc.adjacency_matrix
for col in (5, 777):
    want = orc.resample_adjacency_columns(om, times, nodes, T, 0.35, u, col, col + 1)
    bad = np.nonzero(got[:, col] != want[:, col])[0]
    print("col", col, "mismatches", len(bad), bad[:20])
    ch = np.nonzero(nodes == col + 1)[0]
    for p in bad[:5]:
        # entries of (p, col): children with a parent on node p in their window
        cnt = 0
        for i in ch:
            f = np.searchsorted(times, times[i] - 1.0, side="right")
            cnt += int(np.sum(nodes[f:i] == p + 1))
        print("  p", p, "entries", cnt, "A0", A0[p, col], "got", got[p, col], "want", want[p, col], "u", u[p, col], "W", proc.weights.W[p, col])
