"""MI355X-native hot path for network Hawkes processes (host mirror of
cswaney/NetworkHawkesProcesses.jl's plug-in surface over libnhp.so).

The directory name carries a dot, so it is loaded under the module name `nhp_amd`
(see __graft_entry__.load_package()).  Exports follow src/NetworkHawkesProcesses.jl:36-61;
Julia's `f!` becomes `f_`.
"""
from ._lib import Context, DomainError, NhpError, default_context  # noqa: F401
from .components import (BernoulliNetworkModel, DenseNetworkModel, DenseWeightModel,  # noqa: F401
                         ExponentialImpulseResponse, HomogeneousProcess, LogGaussianCoxProcess,
                         LogitNormalImpulseResponse)
from .continuous import (ContinuousNetworkHawkesProcess, ContinuousStandardHawkesProcess,  # noqa: F401
                         DeviceDataset, device_dataset, intensity, loglikelihood,
                         loglikelihood_gradient, total_intensity)
from .parents import node_counts, parent_counts, resample_parents, uniform_stream  # noqa: F401
from .inference import (MarkovChainMonteCarlo, MaximumLikelihood, logprior, mcmc_, mle_,  # noqa: F401
                        resample_)
from . import synthetic  # noqa: F401
