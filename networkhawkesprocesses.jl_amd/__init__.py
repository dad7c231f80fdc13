"""MI355X-native hot path for network Hawkes processes (host mirror of
cswaney/NetworkHawkesProcesses.jl's plug-in surface over libnhp.so).

The directory name carries a dot, so it is loaded under the module name `nhp_amd`
(see __graft_entry__.load_package()).  Exports follow src/NetworkHawkesProcesses.jl:36-61;
Julia's `f!` becomes `f_`.
"""
from ._lib import Context, DomainError, NhpError, default_context  # noqa: F401
from .components import (BernoulliNetworkModel, DenseNetworkModel, DenseWeightModel, SparseWeightModel,  # noqa: F401
                         ExponentialImpulseResponse, GaussianProcess, HomogeneousProcess,
                         LogGaussianCoxProcess, LogitNormalImpulseResponse, OrnsteinUhlenbeckKernel,
                         PeriodicKernel, SquaredExponentialKernel, split_extract)
from .continuous import (ContinuousHawkesProcess, ContinuousNetworkHawkesProcess,  # noqa: F401
                         ContinuousStandardHawkesProcess, DeviceDataset, HawkesProcess, device_dataset,
                         invalidate_device_datasets, total_intensity)
from . import continuous as _cont
from .discrete import (DiscreteDataset, DiscreteGaussianImpulseResponse, DiscreteHawkesProcess,  # noqa: F401
                       DiscreteHomogeneousProcess, DiscreteLogGaussianCoxProcess, DiscreteNetworkHawkesProcess,
                       DiscreteStandardHawkesProcess, VariationalInference, convolve, disc_parent_counts,
                       disc_resample_adjacency_matrix_,
                       resample_parent_counts, update_, vb_)
from . import discrete as _disc
from .parents import node_counts, parent_counts, resample_parents, uniform_stream  # noqa: F401
from .inference import (MarkovChainMonteCarlo, MaximumLikelihood, logprior,  # noqa: F401
                        resample_adjacency_matrix_)
from . import inference as _inf
from . import synthetic  # noqa: F401
from .synthetic import rand  # noqa: F401   rand(process, duration): the reference's exported simulator name
from . import sharded  # noqa: F401
from .sharded import ShardedDataset, sharded_loglikelihood, sharded_loglikelihood_gradient  # noqa: F401


def loglikelihood(process, data, *args, **kwargs):
    """loglikelihood(process, data; recursive=true) for continuous processes (src/continuous.jl:210,360);
    loglikelihood(process, data[, convolved]) for discrete ones (src/discrete.jl:86-102)."""
    if isinstance(process, DiscreteHawkesProcess):
        return _disc.disc_loglikelihood(process, data, *args, **kwargs)
    return _cont.loglikelihood(process, data, *args, **kwargs)


def intensity(process, data, *args, **kwargs):
    """intensity(process, data, times) for continuous processes (src/continuous.jl:76-96);
    intensity(process, convolved) / intensity(process, data::Matrix) for discrete ones
    (src/discrete.jl:115-131)."""
    if isinstance(process, DiscreteHawkesProcess):
        if isinstance(data, DiscreteDataset) and data.B:
            return _disc.disc_intensity(process, convolved=data, **kwargs)
        return _disc.disc_intensity(process, data, **kwargs)
    return _cont.intensity(process, data, *args, **kwargs)


def mle_(process, data, *args, **kwargs):
    """mle!(process, data; ...) -- src/continuous.jl:144-198 / src/discrete.jl:211-296."""
    if isinstance(process, DiscreteHawkesProcess):
        return _disc.disc_mle_(process, data, *args, **kwargs)
    return _inf.mle_(process, data, *args, **kwargs)


def loglikelihood_gradient(process, data, *args, **kwargs):
    """(ll, gradient) in the order of params(process): continuous [λ0; θ | μ; τ; W], discrete [λ0; vec(W .* θ)]."""
    if isinstance(process, DiscreteHawkesProcess):
        return _disc.disc_loglikelihood_gradient(process, data, *args, **kwargs)
    return _cont.loglikelihood_gradient(process, data, *args, **kwargs)


def mcmc_(process, data, *args, **kwargs):
    """mcmc!(process, data; nsteps, log_freq, verbose) -- src/inference.jl:49-70, continuous or discrete."""
    if isinstance(process, DiscreteHawkesProcess):
        return _disc.disc_mcmc_(process, data, *args, **kwargs)
    return _inf.mcmc_(process, data, *args, **kwargs)


def resample_(process, data, *args, **kwargs):
    """resample!(process, data) (src/continuous.jl:202-208,350-358) / resample!(process, data, convolved)
    (src/discrete.jl:362-368): one Gibbs sweep."""
    if isinstance(process, DiscreteHawkesProcess):
        return _disc.disc_resample_(process, data, *args, **kwargs)
    return _inf.resample_(process, data, *args, **kwargs)



def isstable(process):
    """isstable(process): spectral radius of (A .*) W below one -- src/continuous.jl:58, src/discrete.jl:172,404"""
    return process.isstable()
