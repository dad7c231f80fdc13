"""Discrete-time network process (Bernoulli links): mcmc!.  Mirrors examples/discrete-gaussian-network-hawkes.jl."""
from _common import nhp, np, show


def main(duration=1000, nnodes=2, nbasis=3, nlags=4, dt=1.0, plink=0.5, nsteps=100, seed=0):
    rng = np.random.default_rng(seed)
    baseline = nhp.DiscreteHomogeneousProcess(rng.uniform(size=nnodes), dt)
    impulses = nhp.DiscreteGaussianImpulseResponse(np.ones((nnodes, nnodes, nbasis)) / nbasis, nlags, dt)
    weights = nhp.DenseWeightModel(rng.uniform(size=(nnodes, nnodes)) / nnodes)
    network = nhp.BernoulliNetworkModel(plink, nnodes)
    process = nhp.DiscreteNetworkHawkesProcess(baseline, impulses, weights, network.rand(rng), network, dt)
    print(f"Process is stable? {nhp.isstable(process)}")
    θ = process.params()
    data = nhp.synthetic.rand(process, duration, seed=seed)
    print(f"Generated {data.sum()} events")
    chain = nhp.mcmc_(process, data, nsteps=nsteps, seed=seed)
    show("true vs mcmc mean", θ, np.mean(chain.samples, axis=0))
    return θ, chain


if __name__ == "__main__":
    main()
