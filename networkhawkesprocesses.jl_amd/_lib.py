"""ctypes binding of libnhp.so -- the only way the host mirror reaches the GPU.

There is no CPU fallback: if the HIP library is missing or no device is present, calls
raise.  Signatures mirror include/nhp.h one to one.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnhp.so")

OK, EINVAL, EDOMAIN, ESHAPE, ENOMEM, EHIP, ENOTIMPL = range(7)
BASELINE_HOMOGENEOUS, BASELINE_LGCP = 0, 1
IMPULSE_EXPONENTIAL, IMPULSE_LOGITNORMAL = 0, 1
LL_RECURSIVE = 1
LL_FULL_RECURSION = 2
MAX_SLOTS = 4096

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_vp = C.c_void_p


class DomainError(ValueError):
    """Mirror of Julia's DomainError (src/baselines.jl:100,106,111,116)."""


class NhpError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("baseline_kind", C.c_int32), ("lambda0", _dp),
                ("grid_x", _dp), ("grid_n", C.c_int32), ("impulse_kind", C.c_int32),
                ("theta", _dp), ("mu", _dp), ("tau", _dp), ("dt_max", C.c_double),
                ("W", _dp), ("A", _dp)]


class GibbsPriors(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("alpha0", "beta0", "kappa", "nu", "a", "b", "mu_mu", "kappa_mu")]


class Stats(C.Structure):
    _fields_ = [("cnt0", _dp), ("Mn", _dp), ("Mnm", _dp), ("Xnm", _dp), ("Vnm", _dp)]


_lib = None


def _declare(lib):
    def f(name, res, *args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)

    i32, i64, u64, dbl = C.c_int32, C.c_int64, C.c_uint64, C.c_double
    f("nhp_abi_version", i32)
    f("nhp_ctx_create", i32, i32, C.POINTER(_vp))
    f("nhp_ctx_destroy", None, _vp)
    f("nhp_last_error", C.c_char_p, _vp)
    f("nhp_ctx_synchronize", i32, _vp)
    f("nhp_ctx_timer_start", i32, _vp)
    f("nhp_ctx_timer_stop", i32, _vp, _dp)
    f("nhp_cont_dataset_create", i32, _vp, _dp, _ip, i64, i32, dbl, dbl, C.POINTER(_vp))
    f("nhp_cont_dataset_create_columns", i32, _vp, _dp, _ip, i64, i32, dbl, dbl, i32, i32, C.POINTER(_vp))
    f("nhp_cont_dataset_destroy", None, _vp)
    f("nhp_cont_dataset_pairs", i64, _vp)
    f("nhp_cont_model_create", i32, _vp, C.POINTER(ModelDesc), C.POINTER(_vp))
    f("nhp_cont_model_update", i32, _vp, _vp, C.POINTER(ModelDesc))
    f("nhp_cont_model_set_params", i32, _vp, _vp, _dp, i64)
    f("nhp_cont_model_destroy", None, _vp)
    f("nhp_cont_model_moments_reset", i32, _vp, _vp)
    f("nhp_cont_model_moments_accumulate", i32, _vp, _vp)
    f("nhp_cont_model_moments_fetch", i32, _vp, _vp, _dp, _dp, i64, C.POINTER(i64))
    f("nhp_cont_loglik", i32, _vp, _vp, _vp, i32, _dp)
    f("nhp_cont_loglik_enqueue", i32, _vp, _vp, _vp, i32, i32)
    f("nhp_ctx_fetch", i32, _vp, i32, i32, _dp)
    f("nhp_cont_loglik_batch", i32, _vp, _vp, C.POINTER(_vp), i32, i32, _dp)
    f("nhp_cont_event_intensity", i32, _vp, _vp, _vp, _dp)
    f("nhp_cont_gibbs_step", i32, _vp, _vp, _vp, C.POINTER(GibbsPriors), u64, u64)
    f("nhp_cont_model_get_params", i32, _vp, _vp, _dp, i64)
    f("nhp_cont_lgcp_loglik", i32, _vp, _vp, _ip, _dp, i32, _dp, _dp)
    f("nhp_disc_set_lgcp_baseline", i32, _vp, _vp, _dp, i32, _dp, dbl)
    f("nhp_disc_lgcp_loglik", i32, _vp, _vp, _dp, dbl, _dp)
    f("nhp_disc_loglik_grad", i32, _vp, _vp, _dp, _dp, _dp, dbl, _dp, _dp, i64)
    f("nhp_disc_resample_parents", i32, _vp, _vp, _dp, _dp, _dp, _dp, dbl, u64, u64, _ip)
    f("nhp_disc_gibbs_step", i32, _vp, _vp, _dp, _dp, _dp, _dp, dbl, dbl, dbl, dbl, dbl, dbl, u64, u64)
    f("nhp_disc_resample_adjacency", i32, _vp, _vp, _dp, _dp, _dp, _dp, dbl, _dp, dbl, _dp, u64, u64, _dp)
    f("nhp_cont_resample_adjacency", i32, _vp, _vp, _vp, _dp, dbl, _dp, u64, u64, _dp, _dp)
    f("nhp_probe_math", i32, _vp, i32, _dp, _dp, i64, _dp)
    f("nhp_probe_rate", i32, _vp, i32, i32, i32, _dp)
    f("nhp_probe_gather", i32, _vp, i32, i32, i64, i32, _dp)
    for name, args in (
        ("nhp_cont_loglik_grad", (_vp, _vp, _vp, i32, _dp, _dp, i64)),
        ("nhp_cont_intensity", (_vp, _vp, _vp, _dp, i64, _dp)),
        ("nhp_cont_resample_parents", (_vp, _vp, _vp, _dp, u64, u64, _ip, _ip, C.POINTER(Stats))),
        ("nhp_disc_dataset_create", (_vp, _ip, i32, i64, C.POINTER(_vp))),
        ("nhp_disc_basis", (i32, i32, dbl, _dp)),
        ("nhp_disc_convolve", (_vp, _vp, _dp, i32, i32, _dp)),
        ("nhp_disc_intensity", (_vp, _vp, _dp, _dp, _dp, _dp, dbl, _dp)),
        ("nhp_disc_loglik", (_vp, _vp, _dp, _dp, _dp, _dp, dbl, _dp)),
        ("nhp_disc_vb_step", (_vp, _vp, dbl, dbl, dbl, dbl, dbl, dbl, _dp, _dp, _dp, _dp, _dp)),
        ("nhp_disc_vb_run", (_vp, _vp, dbl, dbl, dbl, dbl, dbl, dbl, i32, _dp, _dp, _dp, _dp, _dp)),
    ):
        if hasattr(lib, name):
            f(name, i32, *args)
    if hasattr(lib, "nhp_disc_dataset_destroy"):
        f("nhp_disc_dataset_destroy", None, _vp)
    f("nhp_uniform_stream", None, u64, u64, i64, _dp)


def lib():
    """Load libnhp.so (in-tree build).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NhpError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # torch ships its own libamdhip64.so.7; when it is already loaded the dynamic linker
        # resolves our dependency to that copy, so torch must be imported first if it is used.
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if "torch" not in sys.modules else C.RTLD_LOCAL)
        _declare(_lib)
    return _lib


def check(rc, ctx=None):
    if rc == OK:
        return
    msg = lib().nhp_last_error(ctx)
    msg = msg.decode() if msg else ""
    if rc == EDOMAIN:
        raise DomainError(msg)
    if rc == ESHAPE:
        raise ValueError(msg or "shape mismatch")
    if rc == ENOTIMPL:
        raise NotImplementedError(msg)
    raise NhpError(f"libnhp status {rc}: {msg}")


def f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def colmajor(a):
    """[p, c(, b)] numpy array -> flat buffer in the reference's column-major order."""
    return None if a is None else np.asfortranarray(np.asarray(a, dtype=np.float64)).ravel(order="K")


class Context:
    """One HIP device + stream (nhp_ctx)."""

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("NHP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.device = device
        h = _vp()
        check(lib().nhp_ctx_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().nhp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib().nhp_ctx_synchronize(self.h), self.h)

    def timer_start(self):
        check(lib().nhp_ctx_timer_start(self.h), self.h)

    def timer_stop(self):
        ms = C.c_double()
        check(lib().nhp_ctx_timer_stop(self.h, C.byref(ms)), self.h)
        return ms.value

    def fetch(self, first, n):
        out = np.empty(n)
        check(lib().nhp_ctx_fetch(self.h, first, n, dptr(out)), self.h)
        return out


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx
