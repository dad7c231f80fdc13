"""Continuous-time standard process, exponential impulse response: simulate, then mle!.
Mirrors examples/continuous-exponential-standard-hawkes.jl of the reference package."""
from _common import nhp, np, show


def main(duration=1000.0, nnodes=2, seed=0):
    rng = np.random.default_rng(seed)
    baseline = nhp.HomogeneousProcess(rng.uniform(size=nnodes))
    weights = nhp.DenseWeightModel(rng.uniform(size=(nnodes, nnodes)) / nnodes)
    impulses = nhp.ExponentialImpulseResponse(rng.uniform(size=(nnodes, nnodes)) + 0.5)
    process = nhp.ContinuousStandardHawkesProcess(baseline, impulses, weights)
    print(f"Process is stable? {nhp.isstable(process)}")
    θ = process.params()
    data = nhp.synthetic.rand(process, duration, seed=seed)
    print(f"Generated {len(data[0])} events")
    res = nhp.mle_(process, data, verbose=False, seed=seed)
    show("true vs mle", θ, res.maximizer)
    return θ, res


if __name__ == "__main__":
    main()
