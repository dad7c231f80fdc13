"""Continuous-time network process (Bernoulli links), logit-normal impulse response: mcmc!.
Mirrors examples/continuous-logit-normal-network-hawkes.jl."""
from _common import nhp, np, show


def main(duration=1000.0, nnodes=2, plink=0.5, nsteps=200, seed=0):
    rng = np.random.default_rng(seed)
    baseline = nhp.HomogeneousProcess(rng.uniform(size=nnodes))
    weights = nhp.DenseWeightModel(rng.uniform(size=(nnodes, nnodes)) / nnodes)
    impulses = nhp.LogitNormalImpulseResponse(rng.uniform(size=(nnodes, nnodes)), rng.uniform(size=(nnodes, nnodes)) + 0.5, 1.0)
    network = nhp.BernoulliNetworkModel(plink, nnodes)
    links = network.rand(rng)
    process = nhp.ContinuousNetworkHawkesProcess(baseline, impulses, weights, links, network)
    print(f"Process is stable? {nhp.isstable(process)}")
    θ = process.params()
    data = nhp.synthetic.rand(process, duration, seed=seed)
    print(f"Generated {len(data[0])} events")
    chain = nhp.mcmc_(process, data, nsteps=nsteps, seed=seed)
    show("true vs mcmc mean", θ, np.mean(chain.samples, axis=0))
    return θ, chain


if __name__ == "__main__":
    main()
