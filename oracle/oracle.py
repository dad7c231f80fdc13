"""ctypes binding of the CPU oracle (oracle/libnhp_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnhp_oracle.so")

MATH_LIBM, MATH_DET, FAST_INTEGRAL = 0, 1, 2
HOMOGENEOUS, LGCP = 0, 1
EXPONENTIAL, LOGITNORMAL = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


class _Model(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("baseline_kind", C.c_int32), ("lambda0", _dp),
                ("grid_x", _dp), ("grid_n", C.c_int32), ("impulse_kind", C.c_int32),
                ("theta", _dp), ("mu", _dp), ("tau", _dp), ("dt_max", C.c_double),
                ("W", _dp), ("A", _dp)]


def build():
    """Compile the oracle with its Makefile (gcc only)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_impulse_exponential.restype = C.c_double
        _lib.orc_impulse_exponential.argtypes = [C.c_double, C.c_double, C.c_int]
        _lib.orc_impulse_logitnormal.restype = C.c_double
        _lib.orc_impulse_logitnormal.argtypes = [C.c_double] * 4 + [C.c_int]
        _lib.orc_linear_integrate.restype = C.c_double
        _lib.orc_disc_loglik.restype = C.c_double
        _lib.orc_digamma.restype = C.c_double
        _lib.orc_digamma.argtypes = [C.c_double]
        _lib.orc_det_exp.restype = C.c_double
        _lib.orc_det_exp.argtypes = [C.c_double]
        _lib.orc_det_log.restype = C.c_double
        _lib.orc_det_log.argtypes = [C.c_double]
        _lib.orc_cont_pair_count.restype = C.c_int64
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__({1: "EINVAL", 2: "EDOMAIN"}.get(code, str(code)))
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _col(a):
    """N x N (or N x N x B) numpy array in [p, c(, b)] indexing -> Julia column-major buffer."""
    return None if a is None else np.asfortranarray(np.asarray(a, dtype=np.float64)).ravel(order="K")


class ContModel:
    """Continuous process parameters; matrices are indexed [parent, child] like the reference."""

    def __init__(self, lambda0, W, theta=None, mu=None, tau=None, dt_max=np.inf, A=None, grid_x=None):
        self.W = np.asarray(W, dtype=np.float64)
        self.N = self.W.shape[0]
        self.theta = None if theta is None else np.asarray(theta, dtype=np.float64)
        self.mu = None if mu is None else np.asarray(mu, dtype=np.float64)
        self.tau = None if tau is None else np.asarray(tau, dtype=np.float64)
        self.A = None if A is None else np.asarray(A, dtype=np.float64)
        self.dt_max = float(dt_max)
        self.grid_x = None if grid_x is None else _f(grid_x)
        self.lambda0 = _f(lambda0)  # homogeneous: [N]; LGCP: [N, G] row per node
        self.impulse_kind = EXPONENTIAL if theta is not None else LOGITNORMAL
        self._keep = [_col(self.W), _col(self.theta), _col(self.mu), _col(self.tau), _col(self.A),
                      self.lambda0.ravel(), self.grid_x]
        k = self._keep
        self.c = _Model(self.N, HOMOGENEOUS if grid_x is None else LGCP, _p(k[5]), _p(k[6]),
                        0 if grid_x is None else len(self.grid_x), self.impulse_kind,
                        _p(k[1]), _p(k[2]), _p(k[3]), self.dt_max, _p(k[0]), _p(k[4]))

    def params_vector(self):
        """params(process) order for the standard process: [λ0; θ | μ; τ; W] (continuous.jl:116-119)."""
        imp = [self._keep[1]] if self.impulse_kind == EXPONENTIAL else [self._keep[2], self._keep[3]]
        return np.concatenate([self.lambda0.ravel()] + imp + [self._keep[0]])


def _data(times, nodes):
    t = _f(times)
    n = np.ascontiguousarray(nodes, dtype=np.int64)
    return t, n, t.ctypes.data_as(_dp), n.ctypes.data_as(_ip), C.c_int64(len(t))


def loglik_windowed(model, times, nodes, duration, flags=0):
    t, n, tp, np_, M = _data(times, nodes)
    out = C.c_double()
    _chk(lib().orc_cont_loglik_windowed(C.byref(model.c), tp, np_, M, C.c_double(duration), flags, C.byref(out)))
    return out.value


def loglik_windowed_mt(model, times, nodes, duration, flags=0, threads=0):
    """The reference's Threads.@threads branch (src/continuous.jl:224-232) on `threads` OpenMP threads (0 = all)."""
    t, n, tp, np_, M = _data(times, nodes)
    out = C.c_double()
    _chk(lib().orc_cont_loglik_windowed_mt(C.byref(model.c), tp, np_, M, C.c_double(duration), flags, int(threads), C.byref(out)))
    return out.value


def max_threads():
    return int(lib().orc_max_threads())


def loglik_recursive(model, times, nodes, duration, flags=0):
    t, n, tp, np_, M = _data(times, nodes)
    out = C.c_double()
    _chk(lib().orc_cont_loglik_recursive(C.byref(model.c), tp, np_, M, C.c_double(duration), flags, C.byref(out)))
    return out.value


def loglik(model, times, nodes, duration, recursive=True, flags=0):
    """loglikelihood(process, data; recursive) dispatch of src/continuous.jl:212-214."""
    if recursive and model.impulse_kind == EXPONENTIAL:
        return loglik_recursive(model, times, nodes, duration, flags)
    return loglik_windowed(model, times, nodes, duration, flags)


def total_intensity(model, times, nodes, i0=0, i1=None, flags=0):
    t, n, tp, np_, M = _data(times, nodes)
    i1 = len(t) if i1 is None else i1
    out = np.empty(i1 - i0)
    _chk(lib().orc_cont_total_intensity(C.byref(model.c), tp, np_, M, C.c_int64(i0), C.c_int64(i1), flags, _p(out)))
    return out


def intensity(model, times, nodes, q, flags=0):
    t, n, tp, np_, M = _data(times, nodes)
    q = _f(np.atleast_1d(q))
    out = np.empty((model.N, len(q)))
    _chk(lib().orc_cont_intensity(C.byref(model.c), tp, np_, M, _p(q), C.c_int64(len(q)), flags, _p(out)))
    return out.T.copy()  # Q x N


def pair_count(times, dt_max):
    t = _f(times)
    return lib().orc_cont_pair_count(_p(t), C.c_int64(len(t)), C.c_double(dt_max))


def uniform_stream(seed, step, M):
    u = np.empty(M)
    lib().orc_uniform_stream(C.c_uint64(seed), C.c_uint64(step), C.c_int64(M), _p(u))
    return u


def resample_parents(model, times, nodes, u, flags=MATH_DET):
    t, n, tp, np_, M = _data(times, nodes)
    u = _f(u)
    parents = np.empty(len(t), dtype=np.int64)
    pnodes = np.empty(len(t), dtype=np.int64)
    _chk(lib().orc_cont_resample_parents(C.byref(model.c), tp, np_, M, _p(u), flags,
                                         parents.ctypes.data_as(_ip), pnodes.ctypes.data_as(_ip)))
    return parents, pnodes


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_ip)


def node_counts(nodes, N):
    n, np_ = _i64(nodes)
    out = np.empty(N)
    lib().orc_node_counts(np_, C.c_int64(len(n)), C.c_int32(N), _p(out))
    return out


def parent_counts(nodes, parentnodes, N):
    n, np_ = _i64(nodes)
    pn, pnp = _i64(parentnodes)
    out = np.empty((N, N))
    lib().orc_parent_counts(np_, pnp, C.c_int64(len(n)), C.c_int32(N), _p(out))
    return out.T.copy()  # [p, c]


def baseline_node_counts(nodes, parentnodes, N):
    n, np_ = _i64(nodes)
    pn, pnp = _i64(parentnodes)
    out = np.empty(N)
    lib().orc_baseline_node_counts(np_, pnp, C.c_int64(len(n)), C.c_int32(N), _p(out))
    return out


def duration_mean(times, nodes, parents, N):
    t = _f(times)
    n, np_ = _i64(nodes)
    pa, pap = _i64(parents)
    out = np.empty((N, N))
    lib().orc_duration_mean(_p(t), np_, pap, C.c_int64(len(t)), C.c_int32(N), _p(out))
    return out.T.copy()


def log_duration_stats(times, nodes, parents, N, dt_max):
    t = _f(times)
    n, np_ = _i64(nodes)
    pa, pap = _i64(parents)
    X = np.empty((N, N))
    V = np.empty((N, N))
    lib().orc_log_duration_stats(_p(t), np_, pap, C.c_int64(len(t)), C.c_int32(N), C.c_double(dt_max), _p(X), _p(V))
    return X.T.copy(), V.T.copy()


def resample_adjacency(model, times, nodes, duration, rho, u):
    """One sweep of resample_adjacency_matrix!; rho, u: N x N arrays indexed [parent, child].  Returns the new A."""
    t, n, tp, np_, M = _data(times, nodes)
    A = _col(model.A).copy()
    r, uu = _col(np.broadcast_to(rho, model.A.shape)), _col(u)
    _chk(lib().orc_cont_resample_adjacency(C.byref(model.c), tp, np_, M, C.c_double(duration), _p(r), _p(uu), _p(A)))
    return A.reshape(model.A.shape, order="F")


def resample_adjacency_columns(model, times, nodes, duration, rho, u, c0, c1):
    """The same sweep for the 0-based columns [c0, c1) only (columns are independent); other columns come back unchanged."""
    t, n, tp, np_, M = _data(times, nodes)
    A = _col(model.A).copy()
    r, uu = _col(np.broadcast_to(rho, model.A.shape)), _col(u)
    _chk(lib().orc_cont_resample_adjacency_columns(C.byref(model.c), tp, np_, M, C.c_double(duration), _p(r), _p(uu), _p(A),
                                                   C.c_int32(c0), C.c_int32(c1)))
    return A.reshape(model.A.shape, order="F")


def lgcp_loglik(times, nodes, parentnodes, N, grid_x, lam):
    """ll[c] of the baseline-attributed events of every node under candidate grid intensities lam [N, G]."""
    t, n, tp, np_, M = _data(times, nodes)
    pn, pnp = _i64(parentnodes)
    x, L = _f(grid_x), _f(np.asarray(lam, dtype=np.float64).reshape(N, -1))
    out = np.empty(N)
    _chk(lib().orc_lgcp_loglik(tp, np_, pnp, M, C.c_int32(N), _p(x), C.c_int32(len(x)), _p(L), _p(out)))
    return out


def loglik_grad(model, times, nodes, duration, recursive=False):
    t, n, tp, np_, M = _data(times, nodes)
    N = model.N
    nb = N if model.grid_x is None else N * len(model.grid_x)
    P = nb + N * N * (2 if model.impulse_kind == EXPONENTIAL else 3)
    g = np.empty(P)
    out = C.c_double()
    _chk(lib().orc_cont_loglik_grad(C.byref(model.c), tp, np_, M, C.c_double(duration), int(recursive), C.byref(out), _p(g)))
    return out.value, g


# ---- discrete

def disc_basis(L, B, dt=1.0):
    phi = np.empty((B, L))
    _chk(lib().orc_disc_basis(C.c_int32(L), C.c_int32(B), C.c_double(dt), _p(phi)))
    return phi.T.copy()  # [L, B]


def disc_convolve(data, phi):
    """data: N x T int64; phi: L x B.  Returns T x N x B."""
    data = np.asarray(data, dtype=np.int64)
    N, T = data.shape
    L, B = phi.shape
    d = np.asfortranarray(data).ravel(order="K")
    ph = np.asfortranarray(phi).ravel(order="K")
    out = np.empty(T * N * B)
    lib().orc_disc_convolve(d.ctypes.data_as(_ip), C.c_int32(N), C.c_int64(T), _p(ph), C.c_int32(L), C.c_int32(B), _p(out))
    return out.reshape((T, N, B), order="F")


def disc_intensity(conv, lambda0, W, theta, dt=1.0, A=None):
    T, N, B = conv.shape
    cv = np.asfortranarray(conv).ravel(order="K")
    out = np.empty(T * N)
    lib().orc_disc_intensity(_p(cv), C.c_int64(T), C.c_int32(N), C.c_int32(B), _p(_f(lambda0)), _p(_col(W)),
                             _p(_col(theta)), _p(_col(A)), C.c_double(dt), _p(out))
    return out.reshape((T, N), order="F")


def disc_resample_parents(data, conv, lambda0, W, theta, dt=1.0, A=None, seed=0, step=0):
    """Parent counts of one discrete Gibbs sweep, [N, 1 + N*B] (column 0 = baseline, 1 + p*B + b)."""
    data = np.asarray(data, dtype=np.int64)
    T, N, B = conv.shape
    d = np.asfortranarray(data).ravel(order="K")
    cv = np.asfortranarray(conv).ravel(order="K")
    out = np.empty(N * (1 + N * B), dtype=np.int64)
    _chk(lib().orc_disc_resample_parents(d.ctypes.data_as(_ip), _p(cv), C.c_int64(T), C.c_int32(N), C.c_int32(B),
                                         _p(_f(lambda0)), _p(_col(W)), _p(_col(theta)), _p(_col(A)), C.c_double(dt),
                                         C.c_uint64(seed), C.c_uint64(step), out.ctypes.data_as(_ip)))
    return out.reshape((N, 1 + N * B), order="F")


def disc_lgcp_intensity(x, lam, dt, times):
    """intensity(p::DiscreteLogGaussianCoxProcess, times) -> len(times) x N; lam is G x N."""
    lam = np.asarray(lam, dtype=np.float64)
    G, N = lam.shape
    tt = _f(times)
    out = np.empty(len(tt) * N)
    _chk(lib().orc_disc_lgcp_intensity(_p(_f(x)), C.c_int32(G), _p(_col(lam)), C.c_int32(N), C.c_double(dt), _p(tt),
                                       C.c_int64(len(tt)), _p(out)))
    return out.reshape((len(tt), N), order="F")


def disc_intensity_b(conv, base_tn, W, theta, dt=1.0, A=None):
    """Process intensity with a per-bin baseline base_tn (T x N)."""
    T, N, B = conv.shape
    cv = np.asfortranarray(conv).ravel(order="K")
    out = np.empty(T * N)
    lib().orc_disc_intensity_b(_p(cv), C.c_int64(T), C.c_int32(N), C.c_int32(B), None, _p(_col(base_tn)), _p(_col(W)),
                               _p(_col(theta)), _p(_col(A)), C.c_double(dt), _p(out))
    return out.reshape((T, N), order="F")


def disc_resample_parents_b(data, conv, base_tn, W, theta, dt=1.0, A=None, seed=0, step=0):
    """Parent counts with a per-bin baseline; returns (counts [N, 1+NB], baseline counts per bin [T, N])."""
    data = np.asarray(data, dtype=np.int64)
    T, N, B = conv.shape
    d = np.asfortranarray(data).ravel(order="K")
    cv = np.asfortranarray(conv).ravel(order="K")
    out = np.empty(N * (1 + N * B), dtype=np.int64)
    bc = np.empty(T * N, dtype=np.int64)
    _chk(lib().orc_disc_resample_parents_b(d.ctypes.data_as(_ip), _p(cv), C.c_int64(T), C.c_int32(N), C.c_int32(B), None,
                                           _p(_col(base_tn)), _p(_col(W)), _p(_col(theta)), _p(_col(A)), C.c_double(dt),
                                           C.c_uint64(seed), C.c_uint64(step), out.ctypes.data_as(_ip), bc.ctypes.data_as(_ip)))
    return out.reshape((N, 1 + N * B), order="F"), bc.reshape((T, N), order="F")


def disc_lgcp_loglik(s0, x, cand, dt):
    """ll[n] of baseline counts s0 (T x N) under candidate grid intensities cand (G x N)."""
    s0 = np.asarray(s0, dtype=np.int64)
    T, N = s0.shape
    cand = np.asarray(cand, dtype=np.float64)
    s = np.asfortranarray(s0).ravel(order="K")
    out = np.empty(N)
    _chk(lib().orc_disc_lgcp_loglik(s.ctypes.data_as(_ip), C.c_int64(T), C.c_int32(N), _p(_f(x)), C.c_int32(cand.shape[0]),
                                    _p(_col(cand)), C.c_double(dt), _p(out)))
    return out


def disc_resample_adjacency(data, conv, lambda0, W, theta, A, rho, u, dt=1.0):
    """One sweep of the discrete resample_adjacency_matrix!; rho, u, A: N x N indexed [parent, child]."""
    data = np.asarray(data, dtype=np.int64)
    T, N, B = conv.shape
    d = np.asfortranarray(data).ravel(order="K")
    cv = np.asfortranarray(conv).ravel(order="K")
    Aw = _col(A).copy()
    r, uu = _col(np.broadcast_to(rho, (N, N))), _col(u)
    _chk(lib().orc_disc_resample_adjacency(d.ctypes.data_as(_ip), _p(cv), C.c_int64(T), C.c_int32(N), C.c_int32(B),
                                           _p(_f(lambda0)), _p(_col(W)), _p(_col(theta)), C.c_double(dt), _p(r), _p(uu), _p(Aw)))
    return Aw.reshape((N, N), order="F")


def disc_loglik(data, lam):
    data = np.asarray(data, dtype=np.int64)
    N, T = data.shape
    d = np.asfortranarray(data).ravel(order="K")
    lm = np.asfortranarray(lam).ravel(order="K")
    return lib().orc_disc_loglik(d.ctypes.data_as(_ip), _p(lm), C.c_int64(T), C.c_int32(N))


def digamma(x):
    return lib().orc_digamma(float(x))


def disc_vb_step(data, conv, dt, alpha0, beta0, kappa, nu, gamma, alpha_v, beta_v, kappa_v, nu_v, gamma_v):
    """Returns updated (alpha_v, beta_v, kappa_v[p,c], nu_v[p,c], gamma_v[p,c,b])."""
    data = np.asarray(data, dtype=np.int64)
    N, T = data.shape
    B = conv.shape[2]
    d = np.asfortranarray(data).ravel(order="K")
    cv = np.asfortranarray(conv).ravel(order="K")
    av, bv = _f(alpha_v).copy(), _f(beta_v).copy()
    kv, nv, gv = _col(kappa_v).copy(), _col(nu_v).copy(), _col(gamma_v).copy()
    _chk(lib().orc_disc_vb_step(d.ctypes.data_as(_ip), _p(cv), C.c_int64(T), C.c_int32(N), C.c_int32(B),
                                C.c_double(dt), C.c_double(alpha0), C.c_double(beta0), C.c_double(kappa),
                                C.c_double(nu), C.c_double(gamma), _p(av), _p(bv), _p(kv), _p(nv), _p(gv)))
    return av, bv, kv.reshape((N, N), order="F"), nv.reshape((N, N), order="F"), gv.reshape((N, N, B), order="F")


def det_exp(x):
    return lib().orc_det_exp(float(x))


def det_log(x):
    return lib().orc_det_log(float(x))
