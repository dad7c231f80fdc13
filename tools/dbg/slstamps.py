"""Phase timeline of k_windowed_slices from a -DNHP_STAMP build (NHP_LIB=... python tools/dbg/slstamps.py): wave 0 of every
workgroup stamps s_memrealtime (100 MHz, the same clock on every XCD) at its start, after the column is staged, after its
slices, after the block sums, after the ticket.  Times in us from the launch's first workgroup start."""
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
t = (st - st[:, 0].min()) / 100.0                   # us
names = ("start", "column staged", "slices done", "block sums done", "ticket drawn")
for k, name in enumerate(names):
    v = t[:, k]
    print(f"{name:18s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
d = np.diff(t, axis=1)
for k, name in enumerate(("column staging", "slices (pair rows)", "log + block sums", "partials + ticket")):
    print(f"phase {name:20s} mean {d[:, k].mean():6.2f}  p10 {np.percentile(d[:, k], 10):6.2f}  p90 {np.percentile(d[:, k], 90):6.2f}")
print("workgroup lifetime mean %.2f us, max %.2f; last ticket at %.2f us after the first start" % ((t[:, 4] - t[:, 0]).mean(), (t[:, 4] - t[:, 0]).max(), t[:, 4].max()))
order = np.argsort(t[:, 0])
print("start time by dispatch order (every 128th workgroup):", " ".join(f"{t[order[i], 0]:.2f}" for i in range(0, n, 128)))
