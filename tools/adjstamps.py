"""Per-column step statistics of k_adj_sweep from a -DNHP_STAMP build: steps, lone-parent steps and their share of the time."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
proc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
for k in range(3):
    nhp.resample_adjacency_matrix_(proc, (times, nodes, T), seed=1, step=k, ctx=ctx)
buf = np.zeros(4 * 1024, dtype=np.uint64)
assert _lib.lib().nhp_debug_adj_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), 4 * 1024) == 0
st = buf.reshape(1024, 4).astype(np.float64)
CLK = 2300.0
print("columns: total %.1f us mean (max %.1f); steps %.0f mean; lone-parent steps %.1f mean taking %.1f us (%.0f cycles each); group steps %.0f cycles each" % (
    st[:, 0].mean() / CLK, st[:, 0].max() / CLK, st[:, 1].mean(), st[:, 2].mean(), st[:, 3].mean() / CLK,
    st[:, 3].sum() / max(1.0, st[:, 2].sum()), (st[:, 0] - st[:, 3]).sum() / (st[:, 1] - st[:, 2]).sum()))
