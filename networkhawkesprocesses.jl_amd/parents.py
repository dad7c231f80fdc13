"""Parent assignment (host mirror of src/parents.jl:1-79 on top of libnhp.so)."""
import ctypes as C

import numpy as np

from . import _lib
from .continuous import device_dataset


def resample_parents(process, data, u=None, seed=0, step=0, with_stats=False, want_parents=True, ctx=None):
    """resample_parents(process, data) -> (parents, parentnodes) -- src/parents.jl:1-23.

    parents[i] is the 1-based index of the sampled parent event (0 = baseline), parentnodes[i]
    its node (0 = baseline).  The reference draws from Julia's task-local RNG, which is not
    reproducible under threads; here the uniform stream is explicit: `u` (one value per event) or
    Philox4x32-10 keyed (seed, step, event index).  With `with_stats` the Gibbs sufficient
    statistics (src/baselines.jl:87-96, src/parents.jl:61-79, src/impulses.jl:84-96,216-252) come
    back from the same call as a dict of [parent, child]-indexed arrays."""
    ctx = ctx or _lib.default_context()
    ds = device_dataset(process, data, ctx)
    model = process.device_model(ctx)
    M, N = len(ds), process.ndims()
    parents = np.empty(M, dtype=np.int64) if want_parents else None
    pnodes = np.empty(M, dtype=np.int64) if want_parents else None
    uu = None if u is None else _lib.f64(u)
    if uu is not None and len(uu) != M:
        raise ValueError("u must hold one uniform per event")
    st, keep = None, {}
    if with_stats:
        keep = {k: np.empty(N if k in ("cnt0", "Mn") else N * N) for k in ("cnt0", "Mn", "Mnm", "Xnm", "Vnm")}
        st = _lib.Stats(*[_lib.dptr(keep[k]) for k in ("cnt0", "Mn", "Mnm", "Xnm", "Vnm")])
    _lib.check(_lib.lib().nhp_cont_resample_parents(
        ctx.h, ds.h, model.h, _lib.dptr(uu), seed, step, _lib.iptr(parents), _lib.iptr(pnodes),
        C.byref(st) if st is not None else None), ctx.h)
    if not with_stats:
        return parents, pnodes
    stats = {k: (v if k in ("cnt0", "Mn") else v.reshape((N, N), order="F")) for k, v in keep.items()}
    return parents, pnodes, stats


def uniform_stream(seed, step, n):
    """The Philox4x32-10 stream the kernel draws from, evaluated on the host."""
    u = np.empty(n)
    _lib.lib().nhp_uniform_stream(seed, step, n, _lib.dptr(u))
    return u


def node_counts(nodes, nnodes):
    """src/parents.jl:61-68"""
    return np.bincount(np.asarray(nodes, dtype=np.int64) - 1, minlength=nnodes).astype(np.float64)


def parent_counts(nodes, parentnodes, nnodes):
    """src/parents.jl:70-79"""
    nodes, parentnodes = np.asarray(nodes, dtype=np.int64), np.asarray(parentnodes, dtype=np.int64)
    cnts = np.zeros((nnodes, nnodes))
    m = parentnodes > 0
    np.add.at(cnts, (parentnodes[m] - 1, nodes[m] - 1), 1.0)
    return cnts
