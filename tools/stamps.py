"""Phase timeline of k_windowed_batch from a -DNHP_STAMP build (s_memtime of wave 0 of every workgroup at: start, columns
staged, pair loop done, block sums done, ticket done).  Usage:
  EXTRA_FLAGS=-DNHP_STAMP BUILD_DIR=.../build_stamp NHP_LIB_OUT=.../libnhp_stamp.so bash csrc/build.sh; NHP_LIB=... python tools/stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
procs = [nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0) for _ in range(8)]
ds = nhp.device_dataset(procs[0], (times, nodes, T), ctx)
models = [p.device_model(ctx) for p in procs]
NB = int(os.environ.get("NB", 8))
arr = (C.c_void_p * NB)(*[models[i % 8].h for i in range(NB)])
out = np.empty(NB)
for _ in range(3):
    _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, NB, 0, _lib.dptr(out)), ctx.h)
n = 1024
buf = np.zeros(8 * n, dtype=np.uint64)
rc = _lib.lib().nhp_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), 8 * n)
assert rc == 0
st = buf.reshape(n, 8)[:, :5].astype(np.int64)
t0 = st[:, 0].min()
CLK = float(os.environ.get("CLK_MHZ", 2300.0))
rel = (st - t0) / CLK             # s_memtime ticks are shader cycles (MI355X_MICROARCH.md); counters of different XCDs are not aligned
print("kernel span (first start -> last end): %.1f us" % rel[:, 4].max())
d = np.diff(st, axis=1) / CLK
for name, col in zip(("staging", "pair loop", "block sums", "ticket"), range(4)):
    print(f"{name:12s} mean {d[:, col].mean():7.2f} us   p10 {np.percentile(d[:, col], 10):7.2f}   p90 {np.percentile(d[:, col], 90):7.2f}")
print("workgroup lifetime mean %.2f us" % ((st[:, 4] - st[:, 0]) / CLK).mean())
starts = np.sort(rel[:, 0])
print("workgroup starts: first 256 by %.1f us, 512 by %.1f, 768 by %.1f, last at %.1f" % (starts[255], starts[511], starts[767], starts[-1]))
