"""Continuous-time standard process, logit-normal impulse response: mle! then mcmc!.
Mirrors examples/continuous-logit-normal-standard-hawkes.jl."""
from _common import nhp, np, show


def main(duration=1000.0, nnodes=2, nsteps=200, seed=0):
    rng = np.random.default_rng(seed)
    baseline = nhp.HomogeneousProcess(rng.uniform(size=nnodes))
    weights = nhp.DenseWeightModel(rng.uniform(size=(nnodes, nnodes)) / nnodes)
    impulses = nhp.LogitNormalImpulseResponse(rng.uniform(size=(nnodes, nnodes)), rng.uniform(size=(nnodes, nnodes)) + 0.5, 1.0)
    process = nhp.ContinuousStandardHawkesProcess(baseline, impulses, weights)
    print(f"Process is stable? {nhp.isstable(process)}")
    θ = process.params()
    data = nhp.synthetic.rand(process, duration, seed=seed)
    print(f"Generated {len(data[0])} events")
    res = nhp.mle_(process, data, seed=seed)
    show("true vs mle", θ, res.maximizer)
    process.params_(θ)                                   # reset parameters
    chain = nhp.mcmc_(process, data, nsteps=nsteps, seed=seed)
    show("true vs mcmc mean", θ, np.mean(chain.samples, axis=0))
    return θ, res, chain


if __name__ == "__main__":
    main()
