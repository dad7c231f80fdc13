#!/usr/bin/env python3
"""Stamps of one step of the discrete adjacency sweep (a -DDADJ_STAMP build, NHP_LIB=...): where the step's time goes.
   NHP_DADJ_STAMPS=/tmp/st.bin NHP_LIB=gpurun_in_stamp.so DG_RATE=0.05 python tools/dadj.py; python tools/dbg/dadjstamps.py /tmp/st.bin [sweep]
   (k_dadj_step by default; `sweep`: the stamp points of the cooperative k_dadj_sweep)"""
import sys
import numpy as np
s = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
t0 = s[:, 0][s[:, 0] > 0].min()
us = lambda v: (v - t0) / 100.0                 # s_memrealtime: 100 MHz
names = ["start", "staged", "loop done", "row stored", "ticket 1", "group row stored", "ticket 2", "decided"]
if len(sys.argv) > 2 and sys.argv[2] == "sweep":
    names = ["step starts", "entries done", "tickets / decision left", "next tables requested", "decision seen", "flips carried", "step ends", "decision announced"]
for k, n in enumerate(names):
    col = s[:, k][s[:, k] > 0]
    if len(col):
        v = us(col)
        print(f"{n:24s} n={len(col):4d}  min {v.min():7.2f}  p50 {np.median(v):7.2f}  max {v.max():7.2f} us")
