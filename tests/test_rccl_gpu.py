"""The RCCL entry points of the C ABI (include/nhp.h, multi-GPU section) on hardware.  A one-GPU box cannot host two
ranks of one RCCL clique, so this runs the real thing with world size 1 -- torch.distributed "nccl" process group,
the library's own communicator created from its 128-byte id (nhp_comm_create -> ncclCommInitRank), collectives on
device pointers -- in a child process (a process group is process-global), and checks that every path that goes
through the communicator gives what the single-GPU call gives.  Two ranks are rehearsed over gloo in
test_sharded_gpu.py / test_chains_gloo.py."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import __graft_entry__ as entry
    nhp = entry.load_package()
    from nhp_amd import _lib, chains, sharded
    from helpers import random_case, rel
    ctx = nhp.default_context()
    comm = _lib.comm_for(ctx)
    assert comm is not None and comm.world == 1 and comm.rank == 0
    # host vectors through the device staging buffer
    x = np.arange(5.0)
    assert np.array_equal(comm.allreduce_sum(x), x) and np.array_equal(comm.allgather(x)[0], x)
    # one evaluation "over all ranks": the device-side all-reduce of the result / of [ll; grad]
    c = random_case(12, 6000, 300.0, "exponential", 1.0, seed=3, nhp=nhp)
    sd = nhp.ShardedDataset(c["proc"], c["data"], ctx)
    for rec in (False, True):
        want = nhp.loglikelihood(c["proc"], c["data"], recursive=rec)
        assert nhp.loglikelihood(c["proc"], sd, recursive=rec) == want
        ll, g = nhp.loglikelihood_gradient(c["proc"], sd, recursive=rec)
        wll, wg = nhp.loglikelihood_gradient(c["proc"], c["data"], recursive=rec)
        # (columns cut into several work items add their gradient parts with atomics: equal to rounding, not bitwise)
        assert ll == wll and np.allclose(g, wg, rtol=1e-12, atol=1e-12)
    # mle! with the optimizer on the device, [ll; grad] all-reduced per evaluation == the same run without a communicator
    m1 = random_case(6, 3000, 200.0, "exponential", 1.0, seed=7, nhp=nhp)
    m2 = random_case(6, 3000, 200.0, "exponential", 1.0, seed=7, nhp=nhp)
    guess = np.random.default_rng(1).uniform(0.2, 0.8, len(m1["proc"].params()))
    # (a few steps: the shard's gradient differs from the whole dataset's in the last bits -- atomics -- and an optimizer
    # amplifies that over hundreds of iterations)
    r1 = nhp.mle_(m1["proc"], nhp.ShardedDataset(m1["proc"], m1["data"], ctx), guess=guess, optimizer="device", max_steps=6)
    r2 = nhp.mle_(m2["proc"], m2["data"], guess=guess, optimizer="device", max_steps=6)
    assert r1.steps == r2.steps == 6 and np.allclose(r1.maximizer, r2.maximizer, rtol=1e-7, atol=1e-10) and rel(r1.maximum, r2.maximum) < 1e-9
    # one network chain through nhp_cont_mcmc_run with the communicator == the same chain without it
    a = random_case(6, 3000, 200.0, "logitnormal", 1.0, network=True, seed=5, nhp=nhp)
    b = random_case(6, 3000, 200.0, "logitnormal", 1.0, network=True, seed=5, nhp=nhp)
    ra = nhp.mcmc_(a["proc"], nhp.ShardedDataset(a["proc"], a["data"], ctx), nsteps=12, seed=9, keep_samples=False, moments=True, burn=2)
    rb = nhp.mcmc_(b["proc"], b["data"], nsteps=12, seed=9, keep_samples=False, moments=True, burn=2)
    assert np.array_equal(ra.samples[-1], rb.samples[-1]) and np.array_equal(ra.mean, rb.mean) and ra.n == rb.n == 10
    # config 5's exchange: chain summaries gathered device to device (nhp_gather_moments)
    def make(k):
        return random_case(5, 2500, 200.0, "logitnormal", 1.0, network=True, seed=41, nhp=nhp)["proc"]
    data = random_case(5, 2500, 200.0, "logitnormal", 1.0, network=True, seed=41, nhp=nhp)["data"]
    out = chains.run_chains(make, data, n_chains=2, nsteps=20, base_seed=5, burn=5)
    for k in range(2):
        res = nhp.mcmc_(make(k), data, nsteps=20, seed=chains.chain_seed(5, k), keep_samples=False, moments=True, burn=5)
        assert out[k]["n"][0] == 15 and np.array_equal(out[k]["mean"], res.mean) and np.array_equal(out[k]["m2"], res.m2)
    dist.destroy_process_group()
    print("RCCL-OK")
''')


def test_rccl_paths_with_a_one_rank_clique():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", NHP_COMM="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=env, capture_output=True, text=True, timeout=900)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_child.log"), "w") as f:
        f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
