"""One log-likelihood / gradient evaluation cut into column shards (second way; sharded.py): the shards'
parts add up to the whole and to the oracle, for both formulations, every impulse / baseline / adjacency combination
and skewed node populations.  One process plays all ranks here (the shards sit on the same GPU one after the other);
the exchange itself is covered by tests/test_sharded_gloo.py."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import rel
from test_shapes_gpu import build

pytestmark = pytest.mark.gpu

CASES = [  # N, M, T, kind, dt_max, network, skew, shards
    (3, 2000, 40.0, "exponential", np.inf, True, False, 3),
    (37, 5000, 300.0, "logitnormal", 1.0, True, True, 4),
    (300, 20000, 800.0, "exponential", 1.0, False, True, 8),
    (257, 9000, 30.0, "logitnormal", 5.0, False, False, 2),
    (128, 40000, 1000.0, "exponential", 0.8, True, False, 5),         # mean window 32: the XCD-aware item layout
]


def shards_of(nhp, proc, data, world):
    return [nhp.ShardedDataset(proc, data, rank=r, world=world) for r in range(world)]


@pytest.mark.parametrize("N,M,T,kind,dt_max,network,skew,world", CASES)
def test_column_shards_add_up_to_the_whole(nhp, orc, N, M, T, kind, dt_max, network, skew, world):
    proc, om, data = build(nhp, orc, N, M, T, kind, dt_max, network, seed=N + M, skew=skew)
    t, n, dur = data
    shards = shards_of(nhp, proc, data, world)
    assert [s.ranges for s in shards] == [shards[0].ranges] * world
    assert shards[0].ranges[0][0] == 0 and shards[0].ranges[-1][1] == N
    for rec in ((False, True) if kind == "exponential" else (False,)):
        whole = nhp.loglikelihood(proc, data, recursive=rec)
        parts = [nhp.loglikelihood(proc, s.local, recursive=rec) for s in shards]
        assert rel(sum(parts), whole) < 1e-12
        want = (orc.loglik_recursive if rec else orc.loglik_windowed)(om, t, n, dur, flags=orc.FAST_INTEGRAL)
        assert rel(sum(parts), want) < 1e-11
        if network:            # the gradient is the mle! objective's: standard processes (src/continuous.jl:144)
            continue
        ll, g = nhp.loglikelihood_gradient(proc, data, recursive=rec)
        gs = [nhp.loglikelihood_gradient(proc, s.local, recursive=rec) for s in shards]
        assert rel(sum(x[0] for x in gs), ll) < 1e-12
        gsum = np.sum([x[1] for x in gs], axis=0)
        assert np.max(np.abs(gsum - g)) <= 1e-12 * np.max(np.abs(g))
        # block-separable: every entry is produced by exactly one shard, the others hold an exact zero there
        nonzero = np.sum([x[1] != 0.0 for x in gs], axis=0)
        assert nonzero.max() <= 1


def test_full_recursion_and_lgcp_baseline_on_shards(nhp, orc, monkeypatch):
    rng = np.random.default_rng(5)
    N, M, T = 24, 6000, 200.0
    times = np.sort(rng.uniform(0, T, M))
    nodes = rng.integers(1, N + 1, M).astype(np.int64)
    x = np.linspace(0.0, T, 21)
    lam = [np.exp(rng.normal(0, 0.3, 21)) * 0.8 for _ in range(N)]
    base = nhp.LogGaussianCoxProcess(x, lam)
    th = rng.uniform(1, 5, (N, N))
    proc = nhp.ContinuousStandardHawkesProcess(base, nhp.ExponentialImpulseResponse(th, 1.0, 1.0, 1.0),
                                               nhp.DenseWeightModel(rng.uniform(0, 1, (N, N)) / N))
    data = (times, nodes, T)
    shards = shards_of(nhp, proc, data, 3)
    monkeypatch.setenv("NHP_REC_WINDOW", "0")                      # the O(M·N) recursion itself
    for rec in (False, True):
        whole = nhp.loglikelihood(proc, data, recursive=rec)
        assert rel(sum(nhp.loglikelihood(proc, s.local, recursive=rec) for s in shards), whole) < 1e-12
        ll, g = nhp.loglikelihood_gradient(proc, data, recursive=rec)
        gs = [nhp.loglikelihood_gradient(proc, s.local, recursive=rec) for s in shards]
        assert rel(sum(v[0] for v in gs), ll) < 1e-12
        assert np.max(np.abs(np.sum([v[1] for v in gs], axis=0) - g)) <= 1e-12 * np.max(np.abs(g))


def test_batch_on_a_shard_and_what_a_shard_refuses(nhp, orc):
    from nhp_amd import _lib
    proc, om, data = build(nhp, orc, 40, 8000, 500.0, "exponential", 1.0, False, seed=11)
    ctx = _lib.default_context()
    shards = shards_of(nhp, proc, data, 2)
    models = [proc.device_model(ctx)]
    arr = (C.c_void_p * 1)(models[0].h)
    tot = 0.0
    for s in shards:
        out = np.empty(1)
        _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, s.local.h, arr, 1, 0, _lib.dptr(out)), ctx.h)
        tot += out[0]
    assert rel(tot, nhp.loglikelihood(proc, data, recursive=False)) < 1e-12
    with pytest.raises(NotImplementedError, match="column shard"):
        nhp.total_intensity(proc, shards[0].local)
    # the parent sampler on a shard: the parents of the children on its nodes, exactly as the whole dataset gives them
    par, pn = nhp.resample_parents(proc, data, seed=1, step=0)[:2]
    par0, pn0 = nhp.resample_parents(proc, shards[0].local, seed=1, step=0)[:2]
    own = (data[1] - 1 >= shards[0].ranges[0][0]) & (data[1] - 1 < shards[0].ranges[0][1])
    assert np.array_equal(par[own], par0[own]) and np.array_equal(pn[own], pn0[own])
    assert not par0[~own].any() and not pn0[~own].any()
    # a query-time intensity table needs no children: allowed
    q = np.array([10.0, 20.0])
    assert np.allclose(nhp.intensity(proc, shards[1].local, q), nhp.intensity(proc, data, q), rtol=1e-13)


def test_bad_column_range_is_rejected(nhp):
    from nhp_amd import _lib
    from nhp_amd.continuous import DeviceDataset
    ctx = _lib.default_context()
    data = (np.array([0.5, 1.0]), np.array([1, 2], dtype=np.int64), 2.0)
    for cols in ((2, 2), (-1, 1), (0, 3), (2, 1)):
        with pytest.raises(nhp.NhpError):
            DeviceDataset(ctx, data, 2, 1.0, columns=cols)


WORKER = '''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import __graft_entry__ as entry
nhp = entry.load_package()
dist.init_process_group("gloo")          # both ranks share the one GPU of the box; on a node it is "nccl", one GPU each
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(2)
N, M, T = 96, 30000, 2000.0
t = np.sort(rng.uniform(0, T, M)); n = rng.integers(1, N + 1, M).astype(np.int64)
def make():
    r = np.random.default_rng(4)
    return nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(r.uniform(0.5, 1.5, N)),
        nhp.ExponentialImpulseResponse(r.uniform(1, 5, (N, N)), 1.0, 1.0, 1.0), nhp.DenseWeightModel(r.uniform(0, 1, (N, N)) / N))
proc, data = make(), (t, n, T)
sd = nhp.ShardedDataset(proc, data)
assert sd.world == 2 and sd.local.columns == sd.ranges[rank]
for rec in (False, True):
    whole = nhp.loglikelihood(proc, data, recursive=rec)
    got = nhp.loglikelihood(proc, sd, recursive=rec)
    assert abs(got - whole) < 1e-12 * abs(whole), (got, whole)
    ll, g = nhp.loglikelihood_gradient(proc, data, recursive=rec)
    ll2, g2 = nhp.loglikelihood_gradient(proc, sd, recursive=rec)
    assert abs(ll2 - ll) < 1e-12 * abs(ll) and np.max(np.abs(g2 - g)) <= 1e-12 * np.max(np.abs(g))
# mle! on the sharded objective: same optimum on both ranks as the single-GPU run from the same start
x0 = np.clip(proc.params() * 1.3, 1e-6, 10.0)
a = nhp.mle_(make(), data, guess=x0, max_steps=5)
b = nhp.mle_(make(), sd, guess=x0, max_steps=5)
assert np.max(np.abs(a.maximizer - b.maximizer)) < 1e-8, np.max(np.abs(a.maximizer - b.maximizer))
both = [None, None]
dist.all_gather_object(both, b.maximizer.tobytes())
assert both[0] == both[1]                  # identical iterates on every rank
dist.barrier()
dist.destroy_process_group()
os.write(1, ("rank %d ok" % rank + chr(10)).encode())
'''


def test_two_ranks_evaluate_one_loglikelihood_together(tmp_path):
    """The N>1 path end to end over the HIP library: two processes, each its column shard, all-reduce between them."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=root))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # NHP_DEVICE=0: both ranks on the one GPU of the test box (the default is LOCAL_RANK, one GPU per rank)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NHP_DEVICE="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-8000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


CHAIN_WORKER = '''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import __graft_entry__ as entry
nhp = entry.load_package()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(6)
N, M, T = 12, 6000, 700.0
t = np.sort(rng.uniform(0, T, M)); n = rng.integers(1, N + 1, M).astype(np.int64)
def make(kind):
    r = np.random.default_rng(9)
    imp = (nhp.ExponentialImpulseResponse(r.uniform(1, 5, (N, N)), 1.0, 1.0, 1.0) if kind == "exp"
           else nhp.LogitNormalImpulseResponse(r.normal(0, 1, (N, N)), r.uniform(0.5, 2, (N, N)), 1.0))
    base, w = nhp.HomogeneousProcess(r.uniform(0.5, 1.5, N)), nhp.DenseWeightModel(r.uniform(0, 1, (N, N)) / N)
    if kind == "exp":
        return nhp.ContinuousStandardHawkesProcess(base, imp, w)
    return nhp.ContinuousNetworkHawkesProcess(base, imp, w, (r.uniform(size=(N, N)) < 0.5).astype(np.float64), nhp.BernoulliNetworkModel(0.5, N))
data = (t, n, T)
for kind in ("exp", "logit"):
    one = make(kind)
    a = nhp.mcmc_(one, data, nsteps=12, seed=4, keep_samples=False, moments=True, burn=3)          # the whole chain on this GPU
    two = make(kind)
    sd = nhp.ShardedDataset(two, data)
    b = nhp.mcmc_(two, sd, nsteps=12, seed=4, keep_samples=False, moments=True, burn=3)            # each rank its columns
    assert np.array_equal(a.samples[-1], b.samples[-1]), (kind, np.max(np.abs(a.samples[-1] - b.samples[-1])))
    assert np.array_equal(a.mean, b.mean) and np.array_equal(a.m2, b.m2) and a.n == b.n == 9
    assert np.array_equal(one.params(), two.params())
    try:
        nhp.mcmc_(two, sd, nsteps=2, seed=4)                 # kept samples would cross PCIe and the ranks every step
        raise SystemExit("expected ValueError")
    except ValueError:
        pass
dist.barrier()
dist.destroy_process_group()
os.write(1, ("rank %d ok" % rank + chr(10)).encode())
'''


def test_two_ranks_run_one_chain_together(tmp_path):
    """ONE mcmc! chain swept by two ranks, each its columns: the chain is the single-GPU chain value for value."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(CHAIN_WORKER.format(root=root))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NHP_DEVICE="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-8000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_gibbs_sweep_on_a_shard_updates_its_columns_only(nhp, orc):
    """Single process: a sweep on a column shard equals the whole-dataset sweep on the shard's columns and leaves the
    other columns of the model untouched."""
    import ctypes as C
    from nhp_amd import _lib, inference
    ctx = _lib.default_context()
    proc, om, data = build(nhp, orc, 10, 5000, 400.0, "logitnormal", 1.0, True, seed=77)
    proc2, _, _ = build(nhp, orc, 10, 5000, 400.0, "logitnormal", 1.0, True, seed=77)
    before = proc.params().copy()
    whole = nhp.device_dataset(proc, data, ctx)
    m1, pri = proc.device_model(ctx), inference._priors(proc)
    _lib.check(_lib.lib().nhp_cont_gibbs_step(ctx.h, whole.h, m1.h, C.byref(pri), 3, 0), ctx.h)
    l1 = inference.resample_adjacency_matrix_(proc, whole, seed=3, step=0, model=m1, ctx=ctx)
    inference._pull_params(proc, m1, ctx)
    sd = nhp.ShardedDataset(proc2, data, rank=1, world=3)
    c0, c1 = sd.ranges[1]
    m2 = proc2.device_model(ctx)
    _lib.check(_lib.lib().nhp_cont_gibbs_step(ctx.h, sd.local.h, m2.h, C.byref(pri), 3, 0), ctx.h)
    l2 = inference.resample_adjacency_matrix_(proc2, sd.local, seed=3, step=0, model=m2, ctx=ctx)
    inference._pull_params(proc2, m2, ctx)
    N = 10
    for name in ("W", "μ", "τ", "A"):
        get = lambda p: (p.adjacency_matrix if name == "A" else getattr(p.weights if name == "W" else p.impulses, name))
        x1, x2 = get(proc), get(proc2)
        assert np.array_equal(x1[:, c0:c1], x2[:, c0:c1]), name
    assert np.array_equal(proc.baseline.λ[c0:c1], proc2.baseline.λ[c0:c1])
    # the other columns keep their initial values
    init, _, _ = build(nhp, orc, 10, 5000, 400.0, "logitnormal", 1.0, True, seed=77)
    keep = np.r_[0:c0, c1:N]
    assert np.array_equal(proc2.weights.W[:, keep], init.weights.W[:, keep])
    assert np.array_equal(proc2.adjacency_matrix[:, keep], init.adjacency_matrix[:, keep])
    assert l2 == proc2.adjacency_matrix[:, c0:c1].sum() and l1 == proc.adjacency_matrix.sum()
