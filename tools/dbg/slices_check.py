#!/usr/bin/env python3
"""Child slices against the pair list / exact records at a given size: python tools/dbg/slices_check.py [N M kbar]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

nhp = entry.load_package()

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
kbar = float(sys.argv[3]) if len(sys.argv) > 3 else 16.0
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=kbar)
proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)


def ll(env, rec=False):
    for k in ("NHP_SLICES", "NHP_PLIST", "NHP_EV8", "NHP_SLICES_CFG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    nhp.invalidate_device_datasets()
    return nhp.loglikelihood(proc, (times, nodes, T), recursive=rec)


ref = ll({"NHP_PLIST": "0", "NHP_EV8": "0"})
print(f"exact   {ref:.9f}")
print(f"pairs   {ll({'NHP_SLICES': '0'}):.9f}")
for cfg in ("64,2", "64,4", "128,2", "128,4", "256,4", "512,4", "1024,4"):
    v = ll({"NHP_SLICES_CFG": cfg})
    print(f"slices {cfg:7s} {v:.9f}  rel {abs(v - ref) / abs(ref):.2e}")
print(f"recursive default {ll({}, True):.9f}   slices off {ll({'NHP_SLICES': '0'}, True):.9f}")
