"""The example scripts (mirrors of the reference's examples/) run end to end on the GPU, and the estimates
land near the truth where the model is well determined."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def run(name, **kw):
    return importlib.import_module(name).main(**kw)


def test_continuous_exponential_mle_recovers_the_parameters():
    θ, res = run("continuous_exponential_standard_hawkes", duration=4000.0)
    assert res.status == "success"
    assert np.all(np.abs(res.maximizer[:2] - θ[:2]) / θ[:2] < 0.25)          # baseline rates
    assert np.max(np.abs(res.maximizer[-4:] - θ[-4:])) < 0.15                # weights


def test_continuous_logit_normal_examples():
    θ, res, chain = run("continuous_logit_normal_standard_hawkes", duration=300.0, nsteps=30)
    assert np.isfinite(res.maximum) and len(chain.samples) == 30
    θ, chain = run("continuous_logit_normal_network_hawkes", duration=300.0, nsteps=30)
    assert len(chain.samples) == 30 and np.all(np.isfinite(chain.samples[-1]))
    θ, res, chain = run("continuous_logit_normal_standard_hawkes_gp", duration=100.0, nsteps=10)
    assert np.isfinite(res.maximum) and len(chain.samples) == 10


def test_discrete_examples():
    θ, res, chain, vb = run("discrete_gaussian_standard_hawkes", duration=3000, nsteps=30)
    assert res.status == "success" and len(chain.samples) == 30
    assert np.all(np.abs(res.maximizer[:2] - θ[:2]) < 0.15)
    θ, chain = run("discrete_gaussian_network_hawkes", duration=1000, nsteps=20)
    assert len(chain.samples) == 20
    θ, res, chain = run("discrete_gaussian_standard_hawkes_gp", duration=1000, nsteps=10)
    assert np.isfinite(res.maximum) and len(chain.samples) == 10
