#!/usr/bin/env python3
"""The device optimizer (nhp_lbfgs_box through nhp_probe_lbfgs) against scipy's L-BFGS-B on separable quadratics
f(x) = ½ Σ h_i (x_i - c_i)²: steps to |f_k - f_{k-1}| < 1e-10 by condition number, with the minimiser inside the box and
with a share of its coordinates beyond the lower bound."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scipy import optimize
import __graft_entry__ as entry
nhp = entry.load_package()
from nhp_amd import _lib
ctx = nhp.Context(0)
fn = _lib.lib().nhp_probe_lbfgs
rng = np.random.default_rng(0)
n = 136
for cond in (1e2, 1e4, 1e6):
    for outside in (0.0, 0.3):
        h = np.exp(rng.uniform(0.0, np.log(cond), n))
        c = rng.uniform(1.0, 5.0, n)
        c[rng.uniform(size=n) < outside] = -1.0                        # the minimiser of these coordinates lies below the box
        x0 = rng.uniform(0.5, 9.0, n)
        x = x0.copy()
        loss, steps, conv, ev = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
        rc = fn(ctx.h, n, _lib.dptr(h), _lib.dptr(c), 1e-6, 10.0, 1e-10, 20000, _lib.dptr(x), C.byref(loss), C.byref(steps), C.byref(conv), C.byref(ev))
        assert rc == 0, rc
        state = {"prev": np.inf, "it": 0}

        def f(z):
            d = z - c
            return 0.5 * np.sum(h * d * d), h * d

        class Stop(Exception):
            pass

        def cb(z):
            v = f(z)[0]
            state["it"] += 1
            if abs(v - state["prev"]) < 1e-10:
                raise Stop
            state["prev"] = v
        try:
            res = optimize.minimize(f, x0, jac=True, method="L-BFGS-B", bounds=[(1e-6, 10.0)] * n, callback=cb,
                                    options=dict(maxiter=20000, ftol=0, gtol=0, maxcor=8))
            fs = res.fun
        except Stop:
            fs = state["prev"]
        xs = np.clip(c, 1e-6, 10.0)
        fopt = f(xs)[0]
        print(f"cond {cond:.0e} outside {outside:.1f}: device {steps.value:5d} steps {ev.value:5d} evals f - f* = {loss.value - fopt:.2e} | scipy (m = 8) {state['it']:5d} steps f - f* = {fs - fopt:.2e}", flush=True)
