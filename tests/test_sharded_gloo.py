"""Column-sharded evaluation of ONE log-likelihood on CPU (second way): the partition of the child nodes,
the column decomposition of the log-likelihood and the all-reduce, with world_size-2 `gloo` processes.  The per-column
GPU evaluation is replaced here by a literal numpy evaluation of one column's part; the sum over ranks must be the
oracle's log-likelihood.  (tests/test_sharded_gpu.py runs the same exchange over the HIP library.)"""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_column_ranges_partition_and_balance(nhp):
    from nhp_amd import sharded
    rng = np.random.default_rng(0)
    for N, world in ((1, 1), (5, 5), (7, 3), (64, 8), (1024, 8), (1000, 6)):
        costs = rng.uniform(0.5, 2.0, N)
        r = sharded.column_ranges(costs, world)
        assert r[0][0] == 0 and r[-1][1] == N and len(r) == world
        assert all(a < b for a, b in r) and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        if N >= 1000:
            parts = np.array([costs[a:b].sum() for a, b in r])
            assert parts.max() / parts.mean() < 1.15
    # one heavy node: still every shard non-empty
    costs = np.ones(16)
    costs[0] = 1e6
    r = sharded.column_ranges(costs, 4)
    assert all(a < b for a, b in r) and r[-1][1] == 16
    with pytest.raises(ValueError):
        sharded.column_ranges(np.ones(3), 4)


def test_column_costs_count_window_pairs(nhp):
    from nhp_amd import sharded
    t = np.array([0.0, 0.5, 1.0, 1.2, 3.0])
    n = np.array([1, 2, 1, 2, 1])
    c = sharded.column_costs(t, n, 2, 1.0)
    # windows (strict t_j > t_i - 1): event 3 (t=1.0) sees 0.5 only -> 1; event 4 (t=1.2) sees 0.5, 1.0 -> 2; event 2 sees 0.0 -> 1
    assert c.tolist() == [1.0 + (0 + 8) + (1 + 8) + (0 + 8), 1.0 + (1 + 8) + (2 + 8)]
    assert sharded.column_costs(t, n, 2, np.inf).tolist() == [1.0 + 8 + (2 + 8) + (4 + 8), 1.0 + (1 + 8) + (3 + 8)]
    assert sharded._all_reduce_sum(np.array([1.5]))[0] == 1.5          # no process group: identity


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, {root!r})
    sys.path.insert(0, os.path.join({root!r}, "tests"))
    import __graft_entry__ as entry
    entry.load_package()
    from nhp_amd import sharded
    from oracle import oracle as orc
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(3)
    N, M, T, dtmax = 9, 700, 60.0, 1.5
    t = np.sort(rng.uniform(0, T, M)); n = rng.integers(1, N + 1, M).astype(np.int64)
    lam0 = rng.uniform(0.5, 1.5, N); W = rng.uniform(0, 1, (N, N)) / N; th = rng.uniform(1, 5, (N, N))
    ranges = sharded.column_ranges(sharded.column_costs(t, n, N, dtmax), world)
    a, b = ranges[rank]
    part = 0.0                                   # this rank's columns, literally (src/continuous.jl:216-237)
    for c in range(a, b):
        part -= lam0[c] * T + sum(W[n[i] - 1, c] for i in range(M))
        for i in np.nonzero(n == c + 1)[0]:
            lam = lam0[c]
            j = i - 1
            while j >= 0 and t[j] > t[i] - dtmax:
                lam += W[n[j] - 1, c] * th[n[j] - 1, c] * np.exp(-th[n[j] - 1, c] * (t[i] - t[j]))
                j -= 1
            part += np.log(lam)
    total = sharded._all_reduce_sum(np.array([part]))[0]
    want = orc.loglik_windowed(orc.ContModel(lam0, W, theta=th, dt_max=dtmax), t, n, T)
    assert abs(total - want) < 1e-9 * abs(want), (total, want)
    g = np.zeros(2 * world); g[2 * rank] = rank + 1.0             # block-separable vectors add to their union
    assert sharded._all_reduce_sum(g).tolist() == [v for r in range(world) for v in (r + 1.0, 0.0)]
    dist.barrier()
    dist.destroy_process_group()
    os.write(1, ("rank %d ok" % rank + chr(10)).encode())
""")


def test_two_rank_gloo_column_shards(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout
