#!/usr/bin/env python3
"""Kernel micro-bench: µs per log-likelihood evaluation (hipEvents on the library stream) for
one workload under the current NHP_GROUP / NHP_CHUNK environment.  Usage:
   python tools/kbench.py windowed_k8 [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

nhp = entry.load_package()
import bench  # noqa: E402


def main():
    name = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    N = int(os.environ.get("KB_N", 1024))
    M = int(os.environ.get("KB_M", 1_000_000))
    ctx = nhp.Context(0)
    r = bench.run_workload(nhp, ctx, name, N, M, steps, 3, lambda: None)
    us = 1e3 * r["dev_ms"] / steps
    B = bench.algorithmic_bytes(N, M, r["kind"])
    print(f"{name:15s} G={os.environ.get('NHP_GROUP','auto'):>4s} chunk={os.environ.get('NHP_CHUNK','auto'):>5s} "
          f"{us:10.1f} us/eval  pairs/s={r['pairs']/us*1e6:.3e}  hbm_frac={B/us/1e3/8000:.4f}  ll={r['ll']:.6f}", flush=True)


if __name__ == "__main__":
    main()
