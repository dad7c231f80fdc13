"""CPU-only checks of the boundary and the host mirror: libnhp.so loads without a GPU and
exports every symbol include/nhp.h declares (no compute call is made), the product never reaches
into oracle/, constructors keep the reference's error behaviour, and the host-side helpers agree
with the reference fixtures."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "networkhawkesprocesses.jl_amd")


@pytest.fixture(scope="module")
def lib():
    path = os.path.join(PKG, "libnhp.so")
    if not os.path.exists(path):
        subprocess.check_call(["bash", os.path.join(PKG, "csrc", "build.sh")])
    return C.CDLL(path)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nhp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nhp_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.nhp_abi_version.restype = C.c_int32
    assert lib.nhp_abi_version() == 2


def test_struct_layouts_match_the_library(lib, nhp):
    """nhp_abi_layout(): sizeof / offsetof of every struct that crosses the boundary, as compiled into libnhp.so.  The
    ctypes Structures of _lib.py are asserted against it at load time (check_layout); the same numbers are written next
    to the Julia structs of julia/NetworkHawkesHIP.jl, which no machine here can execute -- this pins them too."""
    from nhp_amd import _lib
    got = _lib.check_layout(lib)
    # x86-64 / LP64 layout the Julia shim documents: nhp_cont_model_desc 80 bytes, nhp_gibbs_priors 64, nhp_cont_stats 40
    assert got == [80, 0, 4, 8, 16, 24, 28, 32, 40, 48, 56, 64, 72,
                   64, 0, 8, 16, 24, 32, 40, 48, 56,
                   40, 0, 8, 16, 24, 32,
                   4096, 128]
    text = open(os.path.join(PKG, "julia", "NetworkHawkesHIP.jl"), encoding="utf-8").read()
    assert "ABI_LAYOUT" in text
    listed = re.search(r"const ABI_LAYOUT = Int32\[(.*?)\]", text, flags=re.S).group(1)
    assert [int(v) for v in re.findall(r"-?\d+", listed)] == got


def test_rccl_entry_points_fail_cleanly_without_a_communicator(lib):
    """No GPU here: the collectives must refuse a NULL communicator / context with a status, not crash (RCCL itself is only
    dlopen'ed when a communicator is created)."""
    lib.nhp_allreduce_sum.restype = C.c_int32
    x = np.ones(3)
    assert lib.nhp_allreduce_sum(None, None, x.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(3)) == 1     # NHP_EINVAL
    lib.nhp_comm_rank.restype = C.c_int32
    assert lib.nhp_comm_rank(None) == -1
    out = subprocess.run(["ldd", os.path.join(PKG, "libnhp.so")], capture_output=True, text=True).stdout
    assert "rccl" not in out                    # opened on first use, not a load-time dependency


def test_no_gpu_means_a_loud_error_not_a_fallback(lib, nhp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nhp.NhpError):
        nhp.Context(0)
    proc, data = nhp.synthetic.readme_case(seed=0)
    with pytest.raises(nhp.NhpError):
        nhp.loglikelihood(proc, data)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh", ".jl")):
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                hit = re.search(r"\bimport\s+oracle|\bfrom\s+\.*oracle|#include\s+[\"<][^\">]*oracle|"
                                r"libnhp_oracle|nhp_oracle|oracle\.py|orc_[a-z]+\(", text)
                assert hit is None, (dirpath, f, hit.group(0))
    for f in ("libnhp.so",):
        out = subprocess.run(["ldd", os.path.join(PKG, f)], capture_output=True, text=True).stdout
        assert "oracle" not in out


def test_host_uniform_stream_equals_oracle_stream(lib, orc):
    lib.nhp_uniform_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(C.c_double)]
    u = np.empty(4096)
    lib.nhp_uniform_stream(12345, 678, 4096, u.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.array_equal(u, orc.uniform_stream(12345, 678, 4096))
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02


def test_host_basis_equals_oracle_basis(lib, orc):
    lib.nhp_disc_basis.argtypes = [C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_double)]
    for L, B, dt in ((4, 3, 1.0), (32, 8, 0.5), (3, 3, 2.0)):
        phi = np.empty((B, L))
        assert lib.nhp_disc_basis(L, B, dt, phi.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert np.array_equal(phi.T, orc.disc_basis(L, B, dt))


def test_constructor_error_behaviour(nhp):
    # src/baselines.jl:32-34,366-371 and test/baselines.jl:60-67
    with pytest.raises(nhp.DomainError):
        nhp.HomogeneousProcess([1.0, -1.0])
    with pytest.raises(nhp.DomainError):
        nhp.HomogeneousProcess([1.0], 0.0, 1.0)
    for args in ((0.0, 1.0, np.ones(2), np.ones(2), 1.0), (1.0, 0.0, np.ones(2), np.ones(2), 1.0),
                 (1.0, 1.0, [0.0, 1.0], np.ones(2), 1.0), (1.0, 1.0, np.ones(2), [1.0, 0.0], 1.0)):
        with pytest.raises(nhp.DomainError):
            nhp.DiscreteHomogeneousProcess(np.ones(2), *args)
    with pytest.raises(nhp.DomainError):
        nhp.DiscreteHomogeneousProcess(np.ones(2), 0.0)
    with pytest.raises(nhp.DomainError):
        nhp.LogGaussianCoxProcess([0.5, 1.0], [np.ones(2)])       # grid must start at 0 (src/baselines.jl:154)
    with pytest.raises(ValueError):
        nhp.DiscreteGaussianImpulseResponse(np.full((2, 2, 2), 0.3), 4)   # src/impulses.jl:282


def test_discrete_baseline_reference_fixture(nhp):
    # test/baselines.jl:69-88 verbatim
    p = nhp.DiscreteHomogeneousProcess(np.ones(2), 0.5)
    assert np.array_equal(p.intensity(np.arange(0.0, 1.01, 0.1)), 0.5 * np.ones((11, 2)))
    assert p.intensity(1, 0.0) == p.intensity(2, 0.0) == 0.5
    for bad in ((0, 0.0), (3, 0.0), (1, -1.0)):
        with pytest.raises(nhp.DomainError):
            p.intensity(*bad)
    with pytest.raises(nhp.DomainError):
        p.intensity(np.arange(-0.1, 1.0, 0.1))
    data = np.array([[0, 0, 0, 1, 0, 1, 0, 0, 0, 1], [2, 0, 0, 0, 0, 0, 0, 0, 0, 0]])
    Mn, T = p.sufficient_statistics(data)
    assert list(Mn) == [3, 2] and T == 10
    assert np.array_equal(p.integrated_intensity(1.0), [0.5, 0.5])
    assert np.array_equal(p.integrated_intensity(2.0), [1.0, 1.0])
    for bad in ((-1.0,), (0, 1.0), (3, 1.0), (1, -1.0)):
        with pytest.raises(nhp.DomainError):
            p.integrated_intensity(*bad)


def test_params_roundtrip_and_orders(nhp):
    rng = np.random.default_rng(0)
    N = 3
    proc = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(rng.uniform(size=N)),
                                               nhp.LogitNormalImpulseResponse(rng.normal(size=(N, N)), rng.uniform(size=(N, N)), 2.0),
                                               nhp.DenseWeightModel(rng.uniform(size=(N, N))))
    x = proc.params()
    assert len(x) == N + 3 * N * N                                   # [λ0; μ; τ; W]  src/continuous.jl:116-119
    assert np.array_equal(x[N:N + N * N], proc.impulses.μ.ravel(order="F"))
    y = rng.uniform(size=len(x))
    proc.params_(y)
    assert np.array_equal(proc.params(), y)
    with pytest.raises(ValueError):
        proc.params_(y[:-1])                                         # src/impulses.jl:44-45
    net = nhp.ContinuousNetworkHawkesProcess(proc.baseline, proc.impulses, proc.weights, np.ones((N, N)),
                                             nhp.BernoulliNetworkModel(0.5, N))
    assert len(net.params()) == 1 + N + N * N + 2 * N * N + N * N    # [ρ; λ0; W; θ; vec(A)]  :325-333
    assert nhp.parent_counts([1, 1, 2, 2], [0, 1, 0, 2], 2).tolist() == [[1.0, 0.0], [0.0, 1.0]]
    assert nhp.node_counts([1, 1, 2], 3).tolist() == [2.0, 1.0, 0.0]


def test_branching_simulator_statistics(nhp):
    # mean event rate of a stable exponential Hawkes process: (I - Wᵀ)⁻¹ λ0
    lam0, W, th = np.array([0.5, 1.0]), np.array([[0.2, 0.1], [0.3, 0.1]]), np.full((2, 2), 3.0)
    t, n, T = nhp.synthetic.branching_sample(lam0, W, th, 4000.0, seed=0)
    assert np.all(np.diff(t) >= 0) and t.min() >= 0 and t.max() <= T and set(np.unique(n)) == {1, 2}
    want = np.linalg.solve(np.eye(2) - W.T, lam0)
    got = np.bincount(n - 1) / T
    assert np.all(np.abs(got - want) / want < 0.08)


def test_simulators_have_the_right_first_moments(nhp):
    # synthetic.rand: stationary rate of a Hawkes process = (I - Wᵀ)⁻¹ λ0 (continuous), and of the discrete
    # autoregression λ0·dt / (1 - Σ column weights) per bin when the excitation is the same on every link
    W = np.array([[0.2, 0.1], [0.0, 0.3]])
    lam0 = np.array([0.5, 1.0])
    proc = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(lam0), nhp.ExponentialImpulseResponse(2 * np.ones((2, 2))),
                                               nhp.DenseWeightModel(W))
    t, n, T = nhp.synthetic.rand(proc, 4000.0, seed=1)
    assert np.all(np.diff(t) >= 0) and set(np.unique(n)) == {1, 2}
    rate = np.linalg.solve(np.eye(2) - W.T, lam0)
    got = np.bincount(n - 1, minlength=2) / T
    assert np.all(np.abs(got - rate) / rate < 0.08), (got, rate)
    th = np.full((2, 2, 2), 0.5)
    dproc = nhp.DiscreteStandardHawkesProcess.__new__(nhp.DiscreteStandardHawkesProcess)
    dproc.baseline = nhp.DiscreteHomogeneousProcess(np.array([0.2, 0.4]), 1.0)
    dproc.impulses = nhp.DiscreteGaussianImpulseResponse.__new__(nhp.DiscreteGaussianImpulseResponse)
    dproc.impulses.θ, dproc.impulses.nlags, dproc.impulses.dt = th, 4, 1.0
    dproc.impulses.basis = lambda: np.full((4, 2), 0.25)          # unit-mass bases without touching the GPU library
    dproc.weights, dproc.dt = nhp.DenseWeightModel(np.full((2, 2), 0.2)), 1.0
    data = nhp.synthetic.rand(dproc, 20000, seed=2)
    assert data.shape == (2, 20000) and data.min() >= 0
    drate = np.linalg.solve(np.eye(2) - np.full((2, 2), 0.2), np.array([0.2, 0.4]))
    assert np.all(np.abs(data.mean(axis=1) - drate) / drate < 0.08), (data.mean(axis=1), drate)


def test_fast_simulator_matches_the_generation_wise_one(nhp):
    """synthetic.simulated_data merges the N Poisson draws per event into one count plus a categorical child node
    (Poisson splitting): same law as branching_sample, checked on event counts per node and on the mean delay."""
    rng = np.random.default_rng(1)
    N = 5
    lam0, W, th = rng.uniform(0.5, 1.5, N), rng.uniform(0, 1, (N, N)) / N, rng.uniform(1, 5, (N, N))
    p = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(lam0), nhp.ExponentialImpulseResponse(th, 1.0, 1.0, np.inf),
                                            nhp.DenseWeightModel(W))
    T = 20000.0
    a = nhp.synthetic.simulated_data(p, T, seed=3)
    b = nhp.synthetic.branching_sample(lam0, W, th, T, seed=4)
    assert np.all(np.diff(a[0]) >= 0) and a[1].min() >= 1 and a[1].max() <= N and a[2] == T
    ca, cb = np.bincount(a[1], minlength=N + 1)[1:], np.bincount(b[1], minlength=N + 1)[1:]
    want = np.linalg.solve(np.eye(N) - W.T, lam0) * T                    # stationary rates: (I - Wᵀ)⁻¹ λ0
    assert np.all(np.abs(ca - want) < 6 * np.sqrt(want) * 1.5) and np.all(np.abs(cb - want) < 6 * np.sqrt(want) * 1.5)
    with pytest.raises(RuntimeError):
        nhp.synthetic.simulated_data(nhp.ContinuousStandardHawkesProcess(
            nhp.HomogeneousProcess(lam0), nhp.ExponentialImpulseResponse(th, 1.0, 1.0, np.inf), nhp.DenseWeightModel(W * 8)),
            200.0, seed=0, max_events=100_000)


def test_sparse_weight_model_mirror(nhp):
    """SparseWeightModel (src/weights.jl:104-139): constructor defaults, params, and a Gibbs update that is the dense
    one under the present-link prior (κ1, ν1); its variational methods are broken in the reference (D6) and refused."""
    W = np.arange(9.0).reshape(3, 3) / 10
    m = nhp.SparseWeightModel(W, κ1=2.0, ν1=3.0)
    assert (m.κ0, m.ν0, m.κ1, m.ν1) == (1.0, 1.0, 2.0, 3.0) and (m.κ, m.ν) == (2.0, 3.0)
    assert np.array_equal(m.params(), W.ravel(order="F")) and len(m.variational_params()) == 36
    Mn, Mnm = np.array([5.0, 0.0, 2.0]), np.array([[1.0, 0, 2], [0, 0, 0], [3, 1, 0]])
    a = nhp.SparseWeightModel(W, κ1=2.0, ν1=3.0).resample_(Mn, Mnm, np.random.default_rng(1))
    b = nhp.DenseWeightModel(W, 2.0, 3.0).resample_(Mn, Mnm, np.random.default_rng(1))
    assert np.array_equal(a, b)
    proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(np.ones(3), 1.0),
                                             nhp.DiscreteGaussianImpulseResponse(np.full((3, 3, 2), 0.5), 4, 1.0),
                                             nhp.SparseWeightModel(W), 1.0)
    with pytest.raises(NotImplementedError):
        nhp.update_(proc, np.zeros((3, 10), dtype=np.int64), None)
