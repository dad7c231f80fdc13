"""The N>1 path on CPU: world_size-2 `gloo` processes shard independent chains and gather
their summaries with one collective.  The GPU work of a chain is replaced by a
deterministic stand-in so that only the sharding / gather logic is exercised here."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chain_partition(nhp):
    from nhp_amd import chains
    for world in (1, 2, 3, 8):
        owned = [chains.chains_for_rank(8, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(8))
        assert max(map(len, owned)) - min(map(len, owned)) <= 1
    assert len({chains.chain_seed(3, k) for k in range(64)}) == 64
    s = chains.summarize_chain([np.array([1.0, 2.0]), np.array([3.0, 6.0])])
    assert s["mean"].tolist() == [2.0, 4.0] and s["m2"].tolist() == [5.0, 20.0] and s["n"][0] == 2
    assert chains.gather_summaries({0: s}, 1) == {0: s}          # no process group: identity


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, {root!r})
    import __graft_entry__ as entry
    entry.load_package()
    from nhp_amd import chains
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_chains, P = 5, 7
    local = {{}}
    for k in chains.chains_for_rank(n_chains, rank, world):
        rng = np.random.default_rng(chains.chain_seed(11, k))
        local[k] = chains.summarize_chain(rng.normal(k, 1.0, (50, P)))
    allsum = chains.gather_summaries(local, n_chains)
    assert sorted(allsum) == list(range(n_chains))
    for k in range(n_chains):
        rng = np.random.default_rng(chains.chain_seed(11, k))
        want = chains.summarize_chain(rng.normal(k, 1.0, (50, P)))
        assert np.array_equal(allsum[k]["mean"], want["mean"]) and np.array_equal(allsum[k]["m2"], want["m2"])
    dist.barrier()
    dist.destroy_process_group()
    os.write(1, ("rank %d ok" % rank + chr(10)).encode())      # one write: the two ranks share the pipe
""")


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    import socket
    with socket.socket() as sk:                      # a free port: reruns must not collide in TIME_WAIT
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout
