"""Continuous-time standard process with a log Gaussian Cox baseline: mle! then mcmc!.
Mirrors examples/continuous-logit-normal-standard-hawkes-gp.jl."""
from _common import nhp, np, show


def main(duration=100.0, nnodes=2, nsteps_grid=10, nsteps=100, seed=0):
    rng = np.random.default_rng(seed)
    gp = nhp.GaussianProcess(nhp.SquaredExponentialKernel(1.0, 1.0))
    baseline = nhp.LogGaussianCoxProcess.from_gp(gp, 0.0, duration, nsteps_grid, nnodes, rng)
    weights = nhp.DenseWeightModel(rng.uniform(size=(nnodes, nnodes)) / nnodes)
    impulses = nhp.LogitNormalImpulseResponse(rng.uniform(size=(nnodes, nnodes)), rng.uniform(size=(nnodes, nnodes)) + 0.5, 1.0)
    process = nhp.ContinuousStandardHawkesProcess(baseline, impulses, weights)
    print(f"Process is stable? {nhp.isstable(process)}")
    θ = process.params()
    data = nhp.synthetic.rand(process, duration, seed=seed)
    print(f"Generated {len(data[0])} events")
    res = nhp.mle_(process, data, guess=np.clip(θ, 1e-3, 9.0), max_steps=200)
    show("true vs mle", θ, res.maximizer)
    process.params_(θ)
    chain = nhp.mcmc_(process, data, nsteps=nsteps, seed=seed)
    show("true vs mcmc mean", θ, np.mean(chain.samples, axis=0))
    return θ, res, chain


if __name__ == "__main__":
    main()
