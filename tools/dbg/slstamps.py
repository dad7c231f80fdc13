"""Phase timeline of k_windowed_slices from a -DNHP_STAMP build (NHP_LIB=... python tools/dbg/slstamps.py): wave 0 of every
workgroup stamps s_memtime at its start, after the column is staged, after its slices, after the block sums, after the ticket."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
fn = _lib.lib().nhp_debug_stamps_slices
fn.restype = C.c_int
assert fn(buf.ctypes.data_as(C.POINTER(C.c_uint64)), 8 * n) == 0
st = buf.reshape(n, 8)[:, :5].astype(np.int64)
CLK = float(os.environ.get("CLK_MHZ", 2300.0))     # (tools/stamps1.py's calibration: s_memtime ticks are shader cycles; every XCD has its own counter)
d = np.diff(st, axis=1) / CLK
for name, col in zip(("column staging", "slices (pair rows)", "log + block sums", "partials + ticket"), range(4)):
    print(f"{name:22s} mean {d[:, col].mean():7.2f} us   p10 {np.percentile(d[:, col], 10):7.2f}   p90 {np.percentile(d[:, col], 90):7.2f}")
life = (st[:, 4] - st[:, 0]) / CLK
print("workgroup lifetime mean %.2f us, max %.2f" % (life.mean(), life.max()))
for x in range(8):
    sel = np.arange(n) % 8 == x
    s0 = st[sel, 0].min()
    start, end = (st[sel, 0] - s0) / CLK, (st[sel, 4] - s0) / CLK
    print(f"XCD {x}: starts p50 {np.percentile(start, 50):5.2f} max {start.max():5.2f} | ends p10 {np.percentile(end, 10):5.2f} p50 {np.percentile(end, 50):5.2f} max {end.max():5.2f}")
