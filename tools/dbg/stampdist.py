"""Where the headline kernel's workgroups spend the time between the mean lifetime and the kernel's end (a -DNHP_STAMP build):
lifetime against start time, pair count and XCD."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
for _ in range(5):
    nhp.loglikelihood(proc, (times, nodes, T), recursive=False, ctx=ctx)
n = 1024
buf = np.zeros(8 * n, dtype=np.uint64)
assert _lib.lib().nhp_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), 8 * n) == 0
st = buf.reshape(n, 8)[:, :5].astype(np.int64)
CLK = float(os.environ.get("CLK_MHZ", 2300.0))     # (tools/stamps1.py's calibration; every XCD has its own counter: compare inside an XCD)
print("per XCD (workgroups b, b + 8, ...): times in us from the XCD's first start")
tot = []
for x in range(8):
    sel = np.arange(n) % 8 == x
    s0 = st[sel, 0].min()
    start, stage_end, rounds_end, end = [(st[sel, k] - s0) / CLK for k in (0, 1, 2, 3)]
    life = end - start
    cnt = np.bincount(nodes - 1, minlength=N)[:n][sel]
    print(f"XCD {x}: starts p50 {np.percentile(start,50):5.2f} max {start.max():5.2f} | lifetime mean {life.mean():5.2f} p10 {np.percentile(life,10):5.2f} p90 {np.percentile(life,90):5.2f} max {life.max():5.2f}"
          f" | staging {np.mean(stage_end-start):5.2f} rounds {np.mean(rounds_end-stage_end):5.2f} | ends p10 {np.percentile(end,10):5.2f} p50 {np.percentile(end,50):5.2f} max {end.max():5.2f}"
          f" | corr(life,start) {np.corrcoef(life,start)[0,1]:5.2f} corr(life,children) {np.corrcoef(life,cnt)[0,1]:5.2f}")
    tot.append(end.max())
print("last end per XCD:", " ".join(f"{v:.2f}" for v in tot))
