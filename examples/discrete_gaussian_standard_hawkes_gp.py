"""Discrete-time standard process with a discrete log Gaussian Cox baseline: mle! then mcmc!.
Mirrors examples/discrete-gaussian-standard-hawkes-gp.jl."""
from _common import nhp, np, show


def main(duration=1000, nnodes=2, nbasis=3, nlags=4, dt=1.0, nsteps_grid=10, nsteps=50, seed=0):
    rng = np.random.default_rng(seed)
    gp = nhp.GaussianProcess(nhp.SquaredExponentialKernel(1.0, duration / 5.0))
    baseline = nhp.DiscreteLogGaussianCoxProcess.from_gp(gp, -1.0, duration, nsteps_grid, nnodes, dt, rng)
    impulses = nhp.DiscreteGaussianImpulseResponse(np.ones((nnodes, nnodes, nbasis)) / nbasis, nlags, dt)
    weights = nhp.DenseWeightModel(rng.uniform(size=(nnodes, nnodes)) / nnodes)
    process = nhp.DiscreteStandardHawkesProcess(baseline, impulses, weights, dt)
    print(f"Process is stable? {nhp.isstable(process)}")
    θ = process.params()
    data = nhp.synthetic.rand(process, duration, seed=seed)
    print(f"Generated {data.sum()} events")
    res = nhp.mle_(process, data, guess=np.clip(θ, 1e-3, 9.0), max_steps=200)
    show("true vs mle", θ, res.maximizer)
    chain = nhp.mcmc_(process, data, nsteps=nsteps, seed=seed)
    show("true vs mcmc mean", θ, np.mean(chain.samples, axis=0))
    return θ, res, chain


if __name__ == "__main__":
    main()
