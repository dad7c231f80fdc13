"""Discrete-time standard process with Gaussian-basis impulse responses: mle!, mcmc! and vb!.
Mirrors examples/discrete-gaussian-standard-hawkes.jl and -vb.jl."""
from _common import nhp, np, show


def make(nnodes=2, nbasis=3, nlags=4, dt=1.0, seed=0):
    rng = np.random.default_rng(seed)
    baseline = nhp.DiscreteHomogeneousProcess(rng.uniform(size=nnodes), dt)
    impulses = nhp.DiscreteGaussianImpulseResponse(np.ones((nnodes, nnodes, nbasis)) / nbasis, nlags, dt)
    weights = nhp.DenseWeightModel(rng.uniform(size=(nnodes, nnodes)) / nnodes)
    return nhp.DiscreteStandardHawkesProcess(baseline, impulses, weights, dt)


def main(duration=1000, nsteps=100, seed=0):
    process = make(seed=seed)
    print(f"Process is stable? {nhp.isstable(process)}")
    θ = process.params()
    data = nhp.synthetic.rand(process, duration, seed=seed)
    print(f"Generated {data.sum()} events")
    res = nhp.mle_(process, data, seed=seed)
    show("true vs mle", θ, res.maximizer)
    process = make(seed=seed)
    chain = nhp.mcmc_(process, data, nsteps=nsteps, seed=seed)
    show("true vs mcmc mean", θ, np.mean(chain.samples, axis=0))
    process = make(seed=seed)
    vb = nhp.vb_(process, data, max_steps=100)
    print(f"vb: {vb.status} after {vb.step} steps")
    return θ, res, chain, vb


if __name__ == "__main__":
    main()
