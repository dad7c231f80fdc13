import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as entry
nhp = entry.load_package()
from helpers import random_case
for kind, rec, lgcp in [("exponential", True, False), ("exponential", False, False), ("logitnormal", False, False), ("exponential", True, True)]:
    c = random_case(5, 3000, 250.0, kind, 1.5, lgcp=lgcp, seed=31, nhp=nhp)
    guess = np.random.default_rng(5).uniform(0.2, 0.8, len(c["proc"].params()))
    dev = nhp.mle_(c["proc"], c["data"], guess=guess, recursive=rec, f_abstol=1e-9, max_steps=3000, optimizer="device")
    ll, g = nhp.loglikelihood_gradient(c["proc"], c["data"], recursive=rec)
    x = dev.maximizer
    pg = np.where(((x <= 1e-6) & (g < 0)) | ((x >= 10.0) & (g > 0)), 0.0, g)
    c2 = random_case(5, 3000, 250.0, kind, 1.5, lgcp=lgcp, seed=31, nhp=nhp)
    pol = nhp.mle_(c2["proc"], c["data"], guess=x, recursive=rec, f_abstol=1e-9, max_steps=3000)
    c3 = random_case(5, 3000, 250.0, kind, 1.5, lgcp=lgcp, seed=31, nhp=nhp)
    host = nhp.mle_(c3["proc"], c["data"], guess=guess, recursive=rec, f_abstol=1e-9, max_steps=3000)
    c4 = random_case(5, 3000, 250.0, kind, 1.5, lgcp=lgcp, seed=31, nhp=nhp)
    back = nhp.mle_(c4["proc"], c["data"], guess=host.maximizer, recursive=rec, f_abstol=1e-9, max_steps=3000, optimizer="device")
    print(kind, rec, lgcp, "dev", dev.maximum, dev.steps, dev.evaluations, dev.status, "max|pg|", np.abs(pg).max(), "at bound", int(((x<=1e-6)|(x>=10)).sum()),
          "| host polish from dev:", pol.maximum, pol.steps, "| host from guess:", host.maximum, host.steps, "| dev from host opt:", back.maximum, back.steps, flush=True)
