"""Phase timeline of the single-evaluation kernel k_windowed from a -DNHP_STAMP build (see tools/stamps.py)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=float(os.environ.get("KBAR", 8.0)))
proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
for _ in range(5):
    nhp.loglikelihood(proc, (times, nodes, T), recursive=False, ctx=ctx)
n = 1024
buf = np.zeros(8 * n, dtype=np.uint64)
assert _lib.lib().nhp_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), 8 * n) == 0
st = buf.reshape(n, 8)[:, :5].astype(np.int64)
CLK = float(os.environ.get("CLK_MHZ", 2300.0))
d = np.diff(st, axis=1) / CLK
for name, col in zip(("column staging", "rounds (pair loop)", "log + block sums + ticket"), range(3)):
    print(f"{name:28s} mean {d[:, col].mean():7.2f} us   p10 {np.percentile(d[:, col], 10):7.2f}   p90 {np.percentile(d[:, col], 90):7.2f}")
print("workgroup lifetime mean %.2f us, max %.2f" % (((st[:, 3] - st[:, 0]) / CLK).mean(), ((st[:, 3] - st[:, 0]) / CLK).max()))
# start skew within an XCD: blocks b, b+8, ... share an XCD (same counter)
for x in range(2):
    s0 = st[x::8, 0]; e3 = st[x::8, 3]
    print(f"XCD group {x}: starts spread {(s0.max()-s0.min())/CLK:.2f} us, first start -> last end {(e3.max()-s0.min())/CLK:.2f} us")
