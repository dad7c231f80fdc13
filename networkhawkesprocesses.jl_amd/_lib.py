"""ctypes binding of libnhp.so -- the only way the host mirror reaches the GPU.

There is no CPU fallback: if the HIP library is missing or no device is present, calls
raise.  Signatures mirror include/nhp.h one to one.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NHP_LIB: a variant build made by csrc/build.sh with NHP_LIB_OUT (the parameter sweeps in tools/); default: the in-tree library
LIB_PATH = os.environ.get("NHP_LIB") or os.path.join(_HERE, "libnhp.so")
ABI_VERSION = 2

OK, EINVAL, EDOMAIN, ESHAPE, ENOMEM, EHIP, ENOTIMPL, ERCCL = range(8)
COMM_ID_BYTES = 128
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
    f("nhp_abi_layout", i32, C.POINTER(i32), i32)
    u8p = C.POINTER(C.c_uint8)
    f("nhp_comm_unique_id", i32, u8p)
    f("nhp_comm_create", i32, _vp, u8p, i32, i32, C.POINTER(_vp))
    f("nhp_comm_destroy", None, _vp)
    f("nhp_comm_rank", i32, _vp)
    f("nhp_comm_world", i32, _vp)
    f("nhp_allreduce_sum", i32, _vp, _vp, _dp, i64)
    f("nhp_allgather", i32, _vp, _vp, _dp, i64, _dp)
    f("nhp_cont_loglik_allreduce", i32, _vp, _vp, _vp, _vp, i32, _dp)
    f("nhp_cont_loglik_grad_allreduce", i32, _vp, _vp, _vp, _vp, i32, _dp, _dp, i64)
    f("nhp_gather_moments", i32, _vp, _vp, _vp, _dp, _dp, i64, _ip, _dp)
    f("nhp_cont_model_set_rho", i32, _vp, _vp, dbl)
    f("nhp_cont_model_get_rho", i32, _vp, _vp, _dp)
    f("nhp_cont_network_step", i32, _vp, _vp, _vp, _vp, dbl, dbl, u64, u64)
    f("nhp_cont_network_sweep", i32, _vp, _vp, _vp, u64, u64, _dp)
    f("nhp_cont_network_rho", i32, _vp, _vp, dbl, dbl, dbl, dbl, u64, u64)
    f("nhp_cont_mcmc_run", i32, _vp, _vp, _vp, _vp, C.POINTER(GibbsPriors), dbl, dbl, u64, u64, i64, i64)
    f("nhp_disc_mle_run", i32, _vp, _vp, dbl, dbl, dbl, dbl, i32, _dp, i64, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
      C.POINTER(C.c_int32))
    f("nhp_cont_mle_run", i32, _vp, _vp, _vp, _vp, i32, dbl, dbl, dbl, i32, _dp, i64, C.POINTER(C.c_double), C.POINTER(C.c_int32),
      C.POINTER(C.c_int32), C.POINTER(C.c_int32))
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
    f("nhp_cont_model_get_adjacency", i32, _vp, _vp, _dp, i64)
    f("nhp_cont_lgcp_loglik", i32, _vp, _vp, _ip, _dp, i32, _dp, _dp)
    f("nhp_disc_set_lgcp_baseline", i32, _vp, _vp, _dp, i32, _dp, dbl)
    f("nhp_disc_lgcp_loglik", i32, _vp, _vp, _dp, dbl, _dp)
    f("nhp_disc_loglik_grad", i32, _vp, _vp, _dp, _dp, _dp, dbl, _dp, _dp, i64)
    f("nhp_disc_resample_parents", i32, _vp, _vp, _dp, _dp, _dp, _dp, dbl, u64, u64, _ip)
    f("nhp_disc_gibbs_step", i32, _vp, _vp, _dp, _dp, _dp, _dp, dbl, dbl, dbl, dbl, dbl, dbl, u64, u64)
    f("nhp_disc_resample_adjacency", i32, _vp, _vp, _dp, _dp, _dp, _dp, dbl, _dp, dbl, _dp, u64, u64, _dp)
    f("nhp_cont_resample_adjacency", i32, _vp, _vp, _vp, _dp, dbl, _dp, u64, u64, _dp, _dp)
    f("nhp_probe_math", i32, _vp, i32, _dp, _dp, i64, _dp)
    f("nhp_probe_draws", i32, _vp, i32, u64, u64, i64, _dp, _dp, _dp)
    f("nhp_probe_rate", i32, _vp, i32, i32, i32, _dp)
    f("nhp_probe_gather", i32, _vp, i32, i32, i64, i32, _dp)
    f("nhp_probe_stream", i32, _vp, i32, i64, i32, i32, _dp, C.POINTER(C.c_int64))
    f("nhp_probe_lbfgs", i32, _vp, i64, _dp, _dp, dbl, dbl, dbl, i32, _dp, _dp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32))
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
        # torch ships its own libamdhip64.so.7 (and librccl.so.1); whichever copy of a soname is mapped first serves
        # the whole process.  sharded.py / chains.py import torch lazily, so when torch is installed it is imported
        # HERE, before libnhp.so pulls in the system runtime -- one HIP runtime per process, in either import order.
        if "torch" not in sys.modules and os.environ.get("NHP_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if "torch" not in sys.modules else C.RTLD_LOCAL)
        _declare(_lib)
        got = _lib.nhp_abi_version()
        if got != ABI_VERSION:
            raise NhpError(f"{LIB_PATH} has ABI version {got}, this binding expects {ABI_VERSION}: rebuild it")
        check_layout(_lib)
    return _lib


def _layout_of(struct):
    return [C.sizeof(struct)] + [getattr(struct, name).offset for name, _ in struct._fields_]


def check_layout(lib_):
    """The ctypes Structures above against the library's own sizeof / offsetof (nhp_abi_layout)."""
    buf = (C.c_int32 * 64)()
    n = lib_.nhp_abi_layout(buf, 64)
    got = list(buf[:n])
    want = _layout_of(ModelDesc) + _layout_of(GibbsPriors) + _layout_of(Stats) + [MAX_SLOTS, COMM_ID_BYTES]
    if got != want:
        raise NhpError(f"struct layout mismatch between _lib.py and {LIB_PATH}: library {got}, binding {want}")
    return got


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
    if rc == ERCCL:
        raise NhpError(f"RCCL: {msg}")
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


class Comm:
    """nhp_comm: this rank's RCCL communicator (over xGMI), bound to `ctx`.  The library hands RCCL device pointers
    (include/nhp.h, multi-GPU section); the host only carries the 128-byte id from rank 0 to the others -- here through
    the torch.distributed process group that launched the ranks (any backend: it is 128 bytes)."""

    def __init__(self, ctx, rank, world, uid):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        h = _vp()
        check(lib().nhp_comm_create(ctx.h, buf, self.rank, self.world, C.byref(h)), ctx.h)
        self.h = h

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        check(lib().nhp_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_process_group(cls, ctx):
        """Collective over the default torch.distributed group: rank 0 makes the id, everybody joins."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        if dist.get_backend() == "nccl":          # the object broadcast stages on torch's current device: make it ours
            torch.cuda.set_device(ctx.device)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(ctx, rank, world, box[0])

    def close(self):
        if getattr(self, "h", None):
            lib().nhp_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allreduce_sum(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        check(lib().nhp_allreduce_sum(self.ctx.h, self.h, dptr(x), x.size), self.ctx.h)
        return x

    def allgather(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty((self.world,) + x.shape)
        check(lib().nhp_allgather(self.ctx.h, self.h, dptr(x), x.size, dptr(out)), self.ctx.h)
        return out


_comms = {}


def comm_for(ctx):
    """The RCCL communicator of `ctx` over the default torch.distributed group when that group runs on the "nccl"
    backend (one rank per GPU); None otherwise (no group, one rank, or the "gloo" rehearsal on CPU / one shared GPU,
    where two ranks on one device cannot form an RCCL clique)."""
    try:
        import torch.distributed as dist
    except ImportError:                       # pragma: no cover
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size() < 2 and os.environ.get("NHP_COMM") != "rccl":
        return None
    if dist.get_backend() != "nccl" and os.environ.get("NHP_COMM") != "rccl":
        return None
    ctx = ctx or default_context()
    key = id(ctx)
    if key not in _comms:
        _comms[key] = Comm.from_process_group(ctx)
    return _comms[key]
