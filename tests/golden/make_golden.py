#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs + the CPU oracle's outputs for every kernel
family, small enough to commit.  The reference cannot run in this container (Julia absent,
SURVEY.md 8c) and its own tests hold no vectors for this path, so these pin the ORACLE's results
at the time they were generated; `tests/test_golden.py` checks the oracle (CPU) and the HIP path
(GPU) against them.  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import random_case  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def continuous(name, N, M, T, kind, dt_max, network, lgcp, seed):
    c = random_case(N, M, T, kind, dt_max, network=network, lgcp=lgcp, seed=seed, orc=orc)
    om, t, n = c["om"], c["times"], c["nodes"]
    u = orc.uniform_stream(seed, 0, M)
    parents, pnodes = orc.resample_parents(om, t, n, u, flags=orc.MATH_DET)
    q = np.linspace(0.0, T, 9)
    out = dict(times=t, nodes=n, duration=T, dt_max=dt_max, lambda0=om.lambda0, W=om.W,
               ll_windowed=orc.loglik_windowed(om, t, n, T), lam=orc.total_intensity(om, t, n),
               u=u, parents=parents, parentnodes=pnodes, q=q, intensity=orc.intensity(om, t, n, q),
               cnt0=orc.baseline_node_counts(n, pnodes, N), Mnm=orc.parent_counts(n, pnodes, N))
    if kind == "exponential":
        out.update(theta=om.theta, ll_recursive=orc.loglik_recursive(om, t, n, T),
                   Xnm=orc.duration_mean(t, n, parents, N))
    else:
        out.update(mu=om.mu, tau=om.tau)
    if network:
        out["A"] = om.A
    if lgcp:
        out["grid_x"] = om.grid_x
    else:
        ll, g = orc.loglik_grad(om, t, n, T, recursive=False)
        out["grad_windowed"] = g
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def discrete(name, N, T, B, L, seed):
    rng = np.random.default_rng(seed)
    data = rng.poisson(0.3, (N, T)).astype(np.int64)
    W = rng.uniform(0.05, 0.3, (N, N))
    th = rng.dirichlet(np.ones(B), (N, N))
    lam0 = rng.uniform(0.2, 1.0, N)
    phi = orc.disc_basis(L, B, 1.0)
    conv = orc.disc_convolve(data, phi)
    lam = orc.disc_intensity(conv, lam0, W, th, dt=1.0)
    vb = orc.disc_vb_step(data, conv, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, np.ones(N), np.ones(N), np.ones((N, N)),
                          np.ones((N, N)), np.ones((N, N, B)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), data=data, W=W, theta=th, lambda0=lam0, L=L, phi=phi,
                        conv=conv, lam=lam, ll=orc.disc_loglik(data, lam), alpha_v=vb[0], beta_v=vb[1],
                        kappa_v=vb[2], nu_v=vb[3], gamma_v=vb[4])


if __name__ == "__main__":
    continuous("cont_exp_standard", 16, 2000, 120.0, "exponential", 1.0, False, False, 1)
    continuous("cont_exp_network_inf", 4, 600, 60.0, "exponential", np.inf, True, False, 2)
    continuous("cont_logitnormal_network", 16, 2000, 120.0, "logitnormal", 1.0, True, False, 3)
    continuous("cont_logitnormal_lgcp", 6, 800, 50.0, "logitnormal", 2.0, False, True, 4)
    discrete("disc_gaussian_standard", 6, 400, 3, 5, 5)
    print("golden vectors written to", HERE)
