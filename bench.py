#!/usr/bin/env python3
"""bench.py -- log-likelihood evaluations per second at N=1024 nodes, M=1e6 events on MI355X.

One "step" = one complete loglikelihood(process, data) -> scalar (reference:
src/continuous.jl:210-276) with the event data AND the parameters already resident in HBM
(PCIe-inclusive rate: DESIGN.md).  Workloads follow SURVEY.md 8d "S-metric": continuous
exponential standard Hawkes, N=1024, M=1e6, Δtmax=1, mean look-back window K̄ events.

    python bench.py --gpus N --steps K --warmup W [--workload windowed_k8|windowed_k64|
                                                   windowed_k512|recursive|logitnormal_k8]

N>1: launched by torch.distributed.run, one rank per GPU; every rank evaluates its own
independent stream (chains / restarts shard embarrassingly), weak scaling, and the
per-rank results are gathered over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import threading
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EXP_TERM_CEILING = 1.45e12     # exponential pair terms/s on register operands, measured (profiles/README.md)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # AMD spec, fp64 vector; nhp_probe_rate measures 66 TFLOP/s of bare fma
FLOP_PER_EXP_TERM = 44.0       # one pair term w·θ·e^{-θΔt}: 19 fp64 instructions of exp (15 of them fma) + 5 around it = 24
                               # instructions, 20 fma x 2 + 4 = 44 flop (csrc/nhp_math.h)

WORKLOADS = {
    "windowed_k8": dict(kind="exponential", kbar=8.0, recursive=False),
    "windowed_k64": dict(kind="exponential", kbar=64.0, recursive=False),
    "windowed_k512": dict(kind="exponential", kbar=512.0, recursive=False),
    "recursive": dict(kind="exponential", kbar=8.0, recursive=True),                 # default path: truncated window when its bound allows
    "recursive_full": dict(kind="exponential", kbar=8.0, recursive=True, full=True),  # the O(M·N) recursion itself
    "logitnormal_k8": dict(kind="logitnormal", kbar=8.0, recursive=False),
    # SURVEY 8d "realistic" set: the events are drawn from the model itself (children clustered behind their parents,
    # burstier windows: at kbar 8, mean 9.1, s.d. 4.6, max 46 against 8.0 / 2.8 / 27 for the uniform times); M is what
    # the draw gives.  The default run takes the kbar-32 twin (16 lanes per child): its kernel instantiation differs from
    # the headline's k_windowed_pairs<0,4,1,512> (the cached pair list), whose rocprofv3 per-symbol average must stay the headline's alone.
    "simulated_k8": dict(kind="exponential", kbar=8.0, recursive=False, simulated=True),
    "simulated_k32": dict(kind="exponential", kbar=32.0, recursive=False, simulated=True),
}


def algorithmic_bytes(N, M, kind):
    """B_alg = 16·M + 8·P + 8: times f64 + nodes at the reference's Int64 width,
    every parameter once (P = N + 2N² exponential, N + 3N² logit-normal), one scalar out."""
    P = N + (2 if kind == "exponential" else 3) * N * N
    return 16 * M + 8 * P + 8


def measured_traffic(workload):
    """(bytes per launch, commit) of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json,
    written by tools/traffic.sh: FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE; `commit` = the source state
    the passes ran on).  PMC counters cannot be collected inside this process, so the number is a record, and the line says
    which commit it was taken at; (None, None) when the workload has not been profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f)[workload]
        return rec["traffic_bytes_per_launch"], rec.get("commit")
    except (OSError, KeyError, ValueError):
        return None, None


def run_workload(nhp, ctx, name, N, M, steps, warmup, sync):
    import ctypes as C
    from nhp_amd import _lib
    w = WORKLOADS[name]
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=w["kbar"])
    proc = nhp.synthetic.s_metric_process(N, M, T, w["kind"], 1.0)
    if w.get("simulated"):
        times, nodes, T = nhp.synthetic.simulated_data(proc, T, seed=0)
        M = len(times)
    t_ds = time.perf_counter()
    ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
    ctx.synchronize()
    t_ds = time.perf_counter() - t_ds          # bucketing on the host + uploads: once per (data, dt_max)
    model = proc.device_model(ctx)
    flags = (_lib.LL_RECURSIVE if w["recursive"] else 0) | (_lib.LL_FULL_RECURSION if w.get("full") else 0)
    lib = _lib.lib()

    def enqueue(k):
        _lib.check(lib.nhp_cont_loglik_enqueue(ctx.h, ds.h, model.h, flags, k % _lib.MAX_SLOTS), ctx.h)

    t_first = time.perf_counter()              # the first evaluation also builds the derived layouts it uses (pair list)
    enqueue(0)
    ctx.synchronize()
    t_first = time.perf_counter() - t_first
    for k in range(warmup):
        enqueue(k)
    ctx.synchronize()
    sync()
    t0 = time.perf_counter()
    ctx.timer_start()
    for k in range(steps):
        enqueue(k)
    dev_ms = ctx.timer_stop()           # hipEvents on the stream the kernels run on; also drains it
    sync()
    wall = time.perf_counter() - t0
    ll = ctx.fetch(0, 1)[0]
    return dict(name=name, wall=wall, dev_ms=dev_ms, ll=float(ll), pairs=int(ds.pairs), M=M, N=N, dataset_ms=1e3 * t_ds, first_ms=1e3 * t_first,
                kind=w["kind"], data=(times, nodes, T), proc=proc, recursive=w["recursive"])


def two_streams(nhp, base_ctx, r, steps, local):
    """Two independent evaluation streams on ONE GPU (two nhp_ctx = two HIP streams; e.g. two chains or restarts per
    GPU): the fixed per-launch costs of one stream -- launch / completion, column staging, the reduction tail -- run
    under the other stream's pair loops.  Same workload and kernels as the headline; reported beside it, never as
    `value`, because a launch that shares the GPU takes longer (the per-launch roofline is the single-stream one)."""
    from nhp_amd import _lib
    lib = _lib.lib()
    # the leg's launches take another instantiation of the slice kernel (4 rows per request instead of 2: +0.3 us alone), so
    # that rocprofv3's per-symbol average of the headline kernel holds the headline's launches only
    os.environ["NHP_SLICES_CFG"] = "512,4"
    ctxs = [base_ctx, nhp.Context(local)]
    dss, models, keep = [], [], []
    for ctx in ctxs:
        proc = nhp.synthetic.s_metric_process(r["N"], r["M"], r["data"][2], r["kind"], 1.0)
        keep.append(proc)
        dss.append(nhp.continuous.DeviceDataset(ctx, r["data"], r["N"], 1.0))
        models.append(proc.device_model(ctx))

    def go(n):
        for k in range(n):
            for c, d, m in zip(ctxs, dss, models):
                _lib.check(lib.nhp_cont_loglik_enqueue(c.h, d.h, m.h, 0, k % _lib.MAX_SLOTS), c.h)
        for c in ctxs:
            c.synchronize()
    go(5)
    t0 = time.perf_counter()
    go(steps)
    dt = time.perf_counter() - t0
    lls = [float(c.fetch(0, 1)[0]) for c in ctxs]
    os.environ.pop("NHP_SLICES_CFG", None)
    B = algorithmic_bytes(r["N"], r["M"], r["kind"])
    return {"streams": 2, "value": 2 * steps / dt, "unit": "log-likelihood evals/sec", "us_per_evaluation": 1e6 * dt / (2 * steps),
            "aggregate_hbm_frac_on_algorithmic_bytes": 2 * steps * B / dt / 1e9 / HBM_PEAK_GBS, "kernel": "k_windowed_slices<512,4,true,false>",
            "loglik": lls}


def cpu_baseline(r, budget_s=12.0):
    """The oracle (a C restatement of the reference's loops, 1 thread) on a bounded prefix of the
    same workload; evals/s extrapolated linearly in M (both formulations are linear in M)."""
    from oracle import oracle as orc
    times, nodes, T = r["data"]
    proc = r["proc"]
    kw = dict(theta=proc.impulses.θ) if r["kind"] == "exponential" else dict(mu=proc.impulses.μ, tau=proc.impulses.τ)
    om = orc.ContModel(proc.baseline.λ, proc.weights.W, dt_max=1.0, **kw)
    M = len(times)

    def run(m):
        t0 = time.perf_counter()
        orc.loglik(om, times[:m], nodes[:m], T * m / M, recursive=r["recursive"], flags=orc.FAST_INTEGRAL)
        return time.perf_counter() - t0

    probe = min(M, 2000 if r["recursive"] else 50_000)
    tp = run(probe)
    m = int(min(M, max(probe, probe * budget_s / max(tp, 1e-6))))
    runs, spent = [], 0.0
    while not runs or (spent < budget_s and len(runs) < 200):     # ~budget_s of CPU work in total
        runs.append(run(m))
        spent += runs[-1]
    ts = sorted(runs)[len(runs) // 2]
    return dict(value=1.0 / (ts * M / m), unit="log-likelihood evals/sec", cores=1, kind="port",
                sample=f"oracle C restatement, 1 thread, first {m} of {M} events, median of {len(runs)} runs "
                       f"({ts:.3f} s each), scaled linearly to M; row sums of W hoisted (stronger baseline)")


def _cpu_share():
    """Cores this process may actually use: the affinity mask, cut to the cgroup's CPU quota when there is one (a GPU box hands a
    one-GPU job a share of the host -- 128 OpenMP threads on a 16-core share ran 30x slower than 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("NHP_BENCH_THREADS", "64")))


def cpu_baseline_all_cores(r, budget_s=6.0):
    """the reference's `Threads.@threads` branch (src/continuous.jl:224-232) restated with OpenMP, on
    every core the box gives this process; windowed formulation only (the reference's recursion is serial)."""
    from oracle import oracle as orc
    times, nodes, T = r["data"]
    proc = r["proc"]
    kw = dict(theta=proc.impulses.θ) if r["kind"] == "exponential" else dict(mu=proc.impulses.μ, tau=proc.impulses.τ)
    om = orc.ContModel(proc.baseline.λ, proc.weights.W, dt_max=1.0, **kw)
    M = len(times)
    threads = max(1, min(orc.max_threads(), _cpu_share()))

    def run(m):
        t0 = time.perf_counter()
        orc.loglik_windowed_mt(om, times[:m], nodes[:m], T * m / M, flags=orc.FAST_INTEGRAL, threads=threads)
        return time.perf_counter() - t0

    run(min(M, 20_000))                                   # start the thread pool
    m = min(M, 200_000)
    tp = run(m)
    m = int(min(M, max(m, m * 1.0 / max(tp, 1e-6))))      # ~1 s per run
    runs, spent = [], 0.0
    while not runs or (spent < budget_s and len(runs) < 50):
        runs.append(run(m))
        spent += runs[-1]
    ts = sorted(runs)[len(runs) // 2]
    return dict(value=1.0 / (ts * M / m), unit="log-likelihood evals/sec", cores=threads, kind="port",
                sample=f"oracle C restatement of the threaded branch, {threads} OpenMP threads, first {m} of {M} events, "
                       f"median of {len(runs)} runs ({ts:.3f} s each), scaled linearly to M")


def default_dispatch_leg(nhp, ctx, args, sync):
    """The reference's DEFAULT call: loglikelihood(process, data) has recursive=true, and for exponential impulses that is the
    O(M·N) recursion which ignores Δtmax (src/continuous.jl:212-214,241-276).  Two ways to that same value: `recursive`
    = the library's default (the full-history sum through a truncated window whose tail is below 2^-60 of every λ_i, DESIGN
    3.2) and `recursive_full` = the 2·M·N-exponential recursion itself (NHP_LL_FULL_RECURSION).  Each with its own CPU
    baseline (the oracle's literal recursion on a prefix, scaled linearly in M) and an fp64-VALU roofline: this path is
    arithmetic-bound, its algorithmic bytes are the headline's."""
    out = {}
    for name in ("recursive", "recursive_full"):
        steps = max(3, args.steps // 20)
        o = run_workload(nhp, ctx, name, args.nodes, args.events, steps, 2, sync)
        mk = o["dev_ms"] / steps
        entry = {"workload": name, "value": steps / o["wall"], "unit": "log-likelihood evals/sec", "steps": steps, "kernel_ms": mk,
                 "loglik": o["ll"], "hbm_frac_on_algorithmic_bytes": algorithmic_bytes(o["N"], o["M"], o["kind"]) / (mk * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if name == "recursive_full":
            terms = 2.0 * o["M"] * o["N"]                    # the reference's count: 2N exponentials per event
            tf = terms * FLOP_PER_EXP_TERM / (mk * 1e-3) / 1e12
            entry["roofline"] = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / FP64_VALU_PEAK_TFLOPS, "exp_terms_per_s": terms / (mk * 1e-3),
                                 "frac_of_measured_exp_term_ceiling": terms / (mk * 1e-3) / EXP_TERM_CEILING, "traffic": None}
        if not args.no_cpu:
            entry["cpu_baseline"] = cpu_baseline(o, budget_s=2.5)
            entry["speedup_vs_cpu_core"] = entry["value"] / entry["cpu_baseline"]["value"]
        out[name] = entry
    return out


def _eight_models(nhp, ctx, r):
    procs = []
    for q in range(8):
        pq = nhp.synthetic.s_metric_process(r["N"], r["M"], r["data"][2], "exponential", 1.0)
        pq.weights.W = pq.weights.W * (1.0 + 0.01 * q)
        pq.impulses.θ = pq.impulses.θ * (1.0 + 0.005 * q)
        procs.append(pq)
    ds = nhp.device_dataset(procs[0], r["data"], ctx)
    return procs, ds, [pq.device_model(ctx) for pq in procs]


def batch_leg(nhp, ctx, r, args, sync):
    """nhp_cont_loglik_batch: S DISTINCT parameter sets against the headline dataset per call -- what the metric's real
    callers issue (the 2P objective calls of a finite-difference gradient inside mle!, src/continuous.jl:190; restarts;
    chain populations).  Four sets share one pass over the dataset's child slices (k_slices_batch: every pair record
    fetched and decoded once for the four).  Roofline on the batch's OWN algorithmic bytes: the data once per pass, every
    parameter set once: (16·M + 4·8·P) per pass of 4."""
    import ctypes as C
    import numpy as np
    from nhp_amd import _lib
    procs, ds, models = _eight_models(nhp, ctx, r)
    P = r["N"] + 2 * r["N"] * r["N"]
    SETS = 4                                                   # parameter sets per launch (cont_slices.hip)
    t_launch, t_commit = measured_traffic("batch_4_sets")      # bytes per LAUNCH (one pass of 4 sets), from the committed PMC passes
    res = {}
    for nb in (8, 32, 64):
        arr = (C.c_void_p * nb)(*[models[q % 8].h for q in range(nb)])
        outb = np.empty(nb)
        reps = max(3, args.steps // (nb // 2))
        _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, nb, 0, _lib.dptr(outb)), ctx.h)
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, nb, 0, _lib.dptr(outb)), ctx.h)
        sync()
        tb = (time.perf_counter() - t0) / reps
        B = (nb // SETS) * (16 * r["M"] + SETS * 8 * P)        # per call: nb/4 passes of 4 sets
        res[f"sets_{nb}"] = {"value": nb / tb, "unit": "log-likelihood evals/sec", "us_per_evaluation": 1e6 * tb / nb,
                             "calls": reps, "loglik_first_last": [float(outb[0]), float(outb[-1])],
                             "roofline": {"bound": "hbm", "achieved": B / tb / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": B / tb / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": B,
                                          "traffic": None if t_launch is None else (nb // SETS) * t_launch,
                                          "traffic_measured_at_commit": t_commit},
                             "exp_terms_per_s": nb * r["pairs"] / tb,
                             "frac_of_measured_exp_term_ceiling": nb * r["pairs"] / tb / EXP_TERM_CEILING}
    res["note"] = ("wall time per call incl. one result fetch; 4 sets per pass (k_slices_batch: one lane per child, the four "
                   "models' columns in LDS, every pair record fetched once for the four); launches alternate between the "
                   "context's two streams")
    return res


def changing_parameters_leg(nhp, ctx, r, args, sync):
    """SURVEY 8d "report both": the evaluation rate when the parameters CHANGE between evaluations.
    (a) device-updated: 16 different parameter sets resident in HBM, evaluation k takes set k mod 16 (no result re-use,
        268 MB of parameters cycling through the 256 MB Infinity Cache);
    (b) re-uploaded: params!(process, x) through nhp_cont_model_set_params (8·P = 16.8 MB over PCIe) before every
        evaluation -- the PCIe-inclusive rate, never `value`."""
    import numpy as np
    from nhp_amd import _lib
    lib = _lib.lib()
    procs = []
    for q in range(16):
        pq = nhp.synthetic.s_metric_process(r["N"], r["M"], r["data"][2], "exponential", 1.0)
        pq.weights.W = pq.weights.W * (1.0 + 0.01 * q)
        procs.append(pq)
    ds = nhp.device_dataset(procs[0], r["data"], ctx)
    models = [pq.device_model(ctx) for pq in procs]
    steps = args.steps
    for k in range(16):
        _lib.check(lib.nhp_cont_loglik_enqueue(ctx.h, ds.h, models[k].h, 0, k), ctx.h)
    ctx.synchronize()
    sync()
    t0 = time.perf_counter()
    ctx.timer_start()
    for k in range(steps):
        _lib.check(lib.nhp_cont_loglik_enqueue(ctx.h, ds.h, models[k % 16].h, 0, k % 16), ctx.h)
    dev_ms = ctx.timer_stop()
    sync()
    wall = time.perf_counter() - t0
    lls = ctx.fetch(0, 16)
    x = procs[0].params()
    n_up = max(5, steps // 10)
    xs = [x * (1.0 + 1e-3 * k) for k in range(4)]
    ll = np.empty(1)
    models[0].set_params(xs[0])
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(n_up):
        models[0].set_params(xs[k % 4])
        _lib.check(lib.nhp_cont_loglik(ctx.h, ds.h, models[0].h, 0, _lib.dptr(ll)), ctx.h)
    t_up = (time.perf_counter() - t0) / n_up
    return {"rotating_16_resident_parameter_sets": {"value": steps / wall, "unit": "log-likelihood evals/sec", "kernel_ms": dev_ms / steps,
                                                    "distinct_logliks": int(len(set(np.round(lls, 6))))},
            "params_uploaded_before_every_evaluation": {"value": 1.0 / t_up, "unit": "log-likelihood evals/sec", "ms_per_evaluation": 1e3 * t_up,
                                                        "bytes_uploaded_per_evaluation": int(8 * len(x)), "loglik": float(ll[0])}}


def mle_leg(nhp, ctx, args):
    """mle!(process, data) end to end on the metric dataset (src/continuous.jl:144-198): iterations of the device-resident
    optimizer (nhp_cont_mle_run: every objective + gradient evaluation reads parameters the previous iteration wrote ON THE
    DEVICE -- SURVEY 8d's "parameters changing every evaluation, device-updated") next to the host route (scipy L-BFGS-B on
    the same analytic gradient: x up, gradient down and a host-side update of 2.1e6-vectors per objective call)."""
    import numpy as np
    N, M = args.nodes, args.events
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
    out = {"workload": f"mle! on the metric dataset (N={N}, M={M}, exponential, mean window 8), start = the generating parameters "
                       "perturbed by U(0.5, 1.5), box [1e-6, 10]"}
    for name, recursive, opt, steps in (("device_optimizer", False, "device", 100), ("device_optimizer_recursive_objective", True, "device", 30),
                                        ("host_optimizer", False, "L-BFGS-B", 3)):
        proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
        guess = np.clip(proc.params() * np.random.default_rng(9).uniform(0.5, 1.5, len(proc.params())), 1e-6, 10.0)
        nhp.device_dataset(proc, (times, nodes, T), ctx)
        t0 = time.perf_counter()
        res = nhp.mle_(proc, (times, nodes, T), guess=guess, recursive=recursive, f_abstol=1e-12, max_steps=steps, optimizer=opt, ctx=ctx)
        dt = time.perf_counter() - t0
        e = {"steps": res.steps, "ms_per_step": 1e3 * dt / max(1, res.steps), "loglik_reached": res.maximum, "parameters": len(guess)}
        ev = getattr(res, "evaluations", None)
        if ev is not None:
            e["objective_and_gradient_evaluations"] = ev
            e["evaluations_per_sec_parameters_updated_on_the_device"] = ev / dt
        out[name] = e
    out["host_over_device_ms_per_step"] = out["host_optimizer"]["ms_per_step"] / out["device_optimizer"]["ms_per_step"]
    return out


def config_workloads(nhp, ctx, which):
    """BASELINE.json configs[1..3] as secondary measurements (one GPU): wall time per call through
    the host mirror, i.e. including parameter upload and result download."""
    import numpy as np
    out = []

    def timed(fn, reps):
        """median of three blocks of `reps` calls (one warm-up first): wall times of host-involving legs vary run to run"""
        fn()
        ctx.synchronize()
        blocks = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            ctx.synchronize()
            blocks.append((time.perf_counter() - t0) / reps)
        return sorted(blocks)[1]

    if "c2" in which:      # continuous exponential standard Hawkes, N=128, ~1e5 events: ll + mle! gradient
        N, M = 128, 100_000
        # mean window 32 -> 16 lanes per child: a different k_windowed instantiation from the headline's
        # (k_windowed_pairs), so that rocprofv3's per-symbol average of the headline kernel is not mixed with these calls
        times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=32.0)
        proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
        ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
        P = len(proc.params())
        for rec in (True, False):
            t_ll = timed(lambda: nhp.loglikelihood(proc, ds, recursive=rec, ctx=ctx), 20)
            t_g = timed(lambda: nhp.loglikelihood_gradient(proc, ds, recursive=rec, ctx=ctx), 20)
            out.append({"workload": f"c2 N=128 M=1e5 exponential, recursive={rec}", "loglik_ms": 1e3 * t_ll,
                        "loglik_plus_gradient_ms": 1e3 * t_g, "params": P,
                        "reference_gradient_cost_in_loglik_calls": 2 * P})
    if "c3" in which:      # continuous logit-normal network Hawkes, N=1024, ~1e6 events: mcmc! step
        N, M = 1024, 1_000_000
        times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
        proc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
        ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
        step = [0]

        def sampler():
            nhp.resample_parents(proc, ds, seed=1, step=step[0], with_stats=True, want_parents=False, ctx=ctx)
            step[0] += 1
        t_s = timed(sampler, 5)
        rng = np.random.default_rng(0)

        def gibbs():
            nhp.resample_(proc, ds, rng, step=step[0], seed=1, ctx=ctx)
            step[0] += 1
        t_g = timed(gibbs, 3)
        import ctypes as C
        from nhp_amd import _lib, inference
        model, pri = proc.device_model(ctx), inference._priors(proc)

        def device_sweep():
            _lib.check(_lib.lib().nhp_cont_gibbs_step(ctx.h, ds.h, model.h, C.byref(pri), 1, step[0]), ctx.h)
            step[0] += 1
        t_d = timed(device_sweep, 20)

        def adjacency():
            inference.resample_adjacency_matrix_(proc, ds, seed=1, step=step[0], model=model, fetch=False, ctx=ctx)
            step[0] += 1
        t_a = timed(adjacency, 5)
        t_m = timed(lambda: _lib.check(_lib.lib().nhp_cont_model_moments_accumulate(ctx.h, model.h), ctx.h), 20)
        # the whole chain inside the library: sweep + adjacency sweep + ρ ~ Beta on the device + moments, one synchronisation
        # per call (nhp_cont_mcmc_run: the loop body of src/inference.jl:55-62)
        _lib.check(_lib.lib().nhp_cont_model_set_rho(ctx.h, model.h, 0.5), ctx.h)
        run_steps = 50

        def resident_chain():
            _lib.check(_lib.lib().nhp_cont_mcmc_run(ctx.h, None, ds.h, model.h, C.byref(pri), 1.0, 1.0, 1, step[0], run_steps, 0), ctx.h)
            step[0] += run_steps
        t_run = timed(resident_chain, 3) / run_steps
        out.append({"workload": "c3 N=1024 M=1e6 logit-normal network, mcmc! step",
                    "device_gibbs_sweep_ms": 1e3 * t_d, "adjacency_sweep_ms": 1e3 * t_a,
                    "mcmc_steps_per_sec": 1.0 / t_run, "mcmc_step_ms_resident_chain": 1e3 * t_run,
                    "mcmc_steps_per_sec_separate_calls": 1.0 / (t_d + t_a), "sample_moments_on_device_ms": 1e3 * t_m,
                    "params_per_sample": 4 * N * N + N + 1,
                    "parent_sampler_plus_stats_to_host_ms": 1e3 * t_s, "host_draw_gibbs_step_ms": 1e3 * t_g,
                    "pairs": int(ds.pairs)})
    if "c4" in which:      # discrete Gaussian-basis standard Hawkes, N=512, K=8, T=1e5
        N, B, L, T = 512, 8, 32, 100_000
        rng = np.random.default_rng(7)
        data = rng.poisson(0.05, (N, T)).astype(np.int64)
        th = np.asfortranarray(np.full((N, N, B), 1.0 / B))      # column-major like the Julia arrays: no repacking per call
        imp = nhp.DiscreteGaussianImpulseResponse(th, L, 1.0)
        proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(rng.uniform(0.02, 0.08, N), 1.0), imp,
                                                 nhp.DenseWeightModel(np.asfortranarray(rng.uniform(0, 1, (N, N)) / N)), 1.0)
        dsd = nhp.DiscreteDataset(ctx, data)
        t_c = timed(lambda: nhp.convolve(proc, dsd, ctx=ctx), 3)
        t_ll = timed(lambda: nhp.loglikelihood(proc, data, convolved=dsd, ctx=ctx), 5)
        t_vb = timed(lambda: nhp.update_(proc, data, dsd, ctx=ctx, n_steps=10), 2) / 10.0     # 10 resident steps per call
        flop = 2.0 * T * N * N * B
        # discrete Gibbs: parent counts of one sweep, and the adjacency sweep of the network twin
        t_lg = timed(lambda: nhp.loglikelihood_gradient(proc, data, convolved=dsd, ctx=ctx), 3)       # the mle! objective + analytic gradient
        t_pc = timed(lambda: nhp.resample_parent_counts(proc, convolved=dsd, seed=1, step=0, ctx=ctx), 2)
        import copy
        gproc = copy.deepcopy(proc)           # a full resample! (parents + device-side conjugate draws); parameters move
        t_gs = timed(lambda: nhp.resample_(gproc, None, dsd, rng, seed=1, step=0, ctx=ctx), 2)
        net = nhp.DiscreteNetworkHawkesProcess(proc.baseline, imp, proc.weights, (rng.uniform(size=(N, N)) < 0.5).astype(np.float64),
                                               nhp.BernoulliNetworkModel(0.5, N), 1.0)
        t_adj = timed(lambda: nhp.disc_resample_adjacency_matrix_(net, convolved=dsd, seed=1, step=0, ctx=ctx), 2)
        # mle! end to end: 8 steps of the device-resident optimizer (nhp_disc_mle_run), 2 of the host route (scipy L-BFGS-B on
        # the same gradient: parameters up, gradient down and a host-side update of 2.1e6-vectors per objective call)
        guess = np.concatenate([rng.uniform(0.02, 0.08, N), rng.uniform(0.0, 1.0, N * N * B) / (N * B)])
        mle = {}
        for opt, n in (("device", 8), ("L-BFGS-B", 2)):
            mproc = copy.deepcopy(proc)
            t0 = time.perf_counter()
            res = nhp.mle_(mproc, dsd, guess=guess, f_abstol=1e-12, max_steps=n, optimizer=opt, ctx=ctx)
            mle[opt] = 1e3 * (time.perf_counter() - t0) / max(1, res.steps)
        out.append({"workload": "c4 discrete N=512 B=8 L=32 T=1e5", "convolve_ms": 1e3 * t_c, "loglik_ms": 1e3 * t_ll,
                    "loglik_tflops_fp64": flop / t_ll / 1e12, "vb_step_ms": 1e3 * t_vb,
                    "vb_tflops_fp64": 2 * flop / t_vb / 1e12, "mfma_fp64_peak_tflops": 78.6,
                    "loglik_plus_gradient_ms": 1e3 * t_lg,
                    "gibbs_parent_counts_ms": 1e3 * t_pc, "gibbs_step_ms": 1e3 * t_gs, "gibbs_adjacency_sweep_ms": 1e3 * t_adj,
                    "mle_ms_per_step_device_optimizer": mle["device"], "mle_ms_per_step_host_optimizer": mle["L-BFGS-B"],
                    "events": int(data.sum())})
    return out


def sharded_leg(nhp, ctx, N, M, world, names):
    """N>1 only, after the timed region: ONE evaluation cut into column shards over the ranks (sharded.py; SURVEY 8e,
    second way) against the same call on one GPU.  Wall time per call (parameters resident), all-reduce included.
    Every rank runs this (the all-reduce is a collective); returns one entry per workload."""
    out = []
    for name in names:
        w = WORKLOADS[name]
        times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=w["kbar"])
        proc = nhp.synthetic.s_metric_process(N, M, T, w["kind"], 1.0)
        data = (times, nodes, T)
        if w.get("full"):
            os.environ["NHP_REC_WINDOW"] = "0"
        try:
            sd = nhp.ShardedDataset(proc, data, ctx)
            ds = nhp.device_dataset(proc, data, ctx)
            reps = 5 if w["recursive"] else 20
            model = proc.device_model(ctx)            # parameters resident, as in the headline measurement

            def timed(target):
                nhp.loglikelihood(proc, target, recursive=w["recursive"], ctx=ctx, model=model)
                t0 = time.perf_counter()
                for _ in range(reps):
                    v = nhp.loglikelihood(proc, target, recursive=w["recursive"], ctx=ctx, model=model)
                return (time.perf_counter() - t0) / reps, v
            t1, whole = timed(ds)
            ts, parts = timed(sd)
            out.append({"workload": name, "ranks": world, "one_gpu_ms": 1e3 * t1, "sharded_ms": 1e3 * ts,
                        "columns_of_rank0": list(sd.ranges[0]), "rel_diff": abs(parts - whole) / abs(whole)})
        finally:
            os.environ.pop("NHP_REC_WINDOW", None)
    return out


def chains_leg(nhp, ctx, rank, steps, sync):
    """N>1 only (BASELINE config 5): every rank runs its own mcmc! chain of the config-3 model inside the library's chain
    driver (nhp_cont_mcmc_run: parents, statistics, conjugate draws, adjacency sweep, ρ ~ Beta and the posterior moments all
    on the device, chain seed = rank) with no exchange between chains (src/inference.jl:49-70 has no cross-chain term); then
    the ONE exchange of the configuration, the chains' posterior sums all-gathered device to device over RCCL
    (nhp_gather_moments).  Returns this rank's wall time for `steps` steps of its own chain, for the gather, and for
    `steps` steps of ONE chain swept by all ranks together (each rank its columns, link counts all-reduced per step)."""
    import ctypes as C
    import numpy as np
    from nhp_amd import _lib, inference, chains
    lib = _lib.lib()
    N, M = 1024, 1_000_000
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
    proc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
    ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
    model, pri = proc.device_model(ctx), inference._priors(proc)
    seed = chains.chain_seed(1, rank)
    _lib.check(lib.nhp_cont_model_set_rho(ctx.h, model.h, proc.network.ρ), ctx.h)
    _lib.check(lib.nhp_cont_model_moments_reset(ctx.h, model.h), ctx.h)

    def run(k0, n):
        _lib.check(lib.nhp_cont_mcmc_run(ctx.h, None, ds.h, model.h, C.byref(pri), proc.network.α, proc.network.β, seed, k0, n, 0), ctx.h)
    run(0, 3)
    sync()
    t0 = time.perf_counter()
    run(3, steps)
    sync()
    wall = time.perf_counter() - t0
    # the two parts that exchange are guarded separately (-1 = failed, reported as null): the chains' own rate above stands
    t_gather = -1.0
    lib_world, lib_rank = 0, -1                     # what the LIBRARY's communicator says (0 / -1: none, the gloo rehearsal)
    try:
        comm = _lib.comm_for(ctx)                   # RCCL communicator of this rank ("nccl" groups); None in the gloo rehearsal
        t_gather = 0.0
        if comm is not None:
            lib_world, lib_rank = int(lib.nhp_comm_world(comm.h)), int(lib.nhp_comm_rank(comm.h))
            L = inference.moments_length(proc)
            s, q = np.empty((comm.world, L)), np.empty((comm.world, L))
            counts, rho = np.empty(comm.world, dtype=np.int64), np.empty((comm.world, 3))
            sync()
            t0 = time.perf_counter()
            _lib.check(lib.nhp_gather_moments(ctx.h, comm.h, model.h, _lib.dptr(s), _lib.dptr(q), L, _lib.iptr(counts), _lib.dptr(rho)), ctx.h)
            sync()
            t_gather = time.perf_counter() - t0
            assert np.all(counts == steps + 3)
    except Exception as exc:
        t_gather = -1.0
        print(f"[rank {rank}] gather of the chains' moments failed: {exc!r}", file=sys.stderr, flush=True)
    # the same chain swept by ALL ranks, each its columns (mcmc_ on a ShardedDataset -> nhp_cont_mcmc_run with the
    # communicator; through the host when the ranks cannot form an RCCL clique)
    t_one = -1.0
    try:
        from nhp_amd.sharded import ShardedDataset
        sproc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
        sd = ShardedDataset(sproc, (times, nodes, T), ctx)
        nhp.mcmc_(sproc, sd, nsteps=3, seed=1, keep_samples=False, moments=True)
        sync()
        t0 = time.perf_counter()
        nhp.mcmc_(sproc, sd, nsteps=steps, seed=1, keep_samples=False, moments=True)
        sync()
        t_one = time.perf_counter() - t0
    except Exception as exc:
        print(f"[rank {rank}] one chain over all ranks failed: {exc!r}", file=sys.stderr, flush=True)
    return wall, t_one, t_gather, float(lib_world), float(lib_rank)


def find_errors(obj, path=""):
    """Paths of every "error" key inside the line: a leg that raised is recorded where it belongs AND named at the top."""
    found = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            if k == "error":
                found.append(path or "/")
            else:
                found += find_errors(v, f"{path}/{k}")
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            found += find_errors(v, f"{path}[{i}]")
    return found


def finish_line(out):
    """`status`: "ok", or "partial" with the paths of the legs that failed (the headline itself is always measured)."""
    bad = find_errors(out)
    out["status"] = "partial" if bad else "ok"
    if bad:
        out["failed_legs"] = bad
    return json.dumps(out)


def headline(args, r, world, wall_s, lls):
    """The contract's JSON line as a dict: the headline measurement (everything else is added to it)."""
    B = algorithmic_bytes(r["N"], r["M"], r["kind"])
    ms_kernel = r["dev_ms"] / args.steps
    out = {
        "metric": "log-likelihood evals/sec (N=1024, M=1e6)",
        "value": world * args.steps / wall_s,
        "unit": "log-likelihood evals/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * wall_s / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"continuous exponential standard Hawkes, N={r['N']}, M={r['M']}, "
                               f"dt_max=1, {args.workload} (S-metric, SURVEY 8d)",
                   "pairs_per_eval": r["pairs"], "independent_streams": world,
                   "dataset_setup_ms_once": r["dataset_ms"], "first_evaluation_ms_incl_layout_build": r["first_ms"],
                   "data_layout": "per dataset, made once from the events (data only, no parameter in it): children bucketed by node, "
                                  "parent-child pairs as child slices (64 children a wavefront, one lane per child, 6-byte records "
                                  "node | delay, row r = every lane's r-th most recent parent); every evaluation computes every pair "
                                  "term from the parameters it is given (see parameters_changing_every_evaluation)"},
        "roofline": {"bound": "hbm", "achieved": B / (ms_kernel * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": B / (ms_kernel * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": measured_traffic(args.workload)[0], "traffic_measured_at_commit": measured_traffic(args.workload)[1],
                     "algorithmic_bytes": B, "kernel_ms": ms_kernel,
                     "pair_rate_per_s": r["pairs"] / (ms_kernel * 1e-3)},
        "loglik": [float(v) for v in lls.cpu()],
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("NHP_BENCH_WORKLOAD", "windowed_k8"), choices=sorted(WORKLOADS))
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--events", type=int, default=1_000_000)
    ap.add_argument("--extra", default=os.environ.get("NHP_BENCH_EXTRA", "windowed_k64,windowed_k512,simulated_k32,logitnormal_k8"),
                    help="comma list of secondary workloads reported under 'other_workloads' (N=1 only)")
    ap.add_argument("--configs", default=os.environ.get("NHP_BENCH_CONFIGS", "c2,c3,c4"),
                    help="comma list of BASELINE configs measured as secondary workloads (N=1 only); '' to skip")
    ap.add_argument("--sharded", default=os.environ.get("NHP_BENCH_SHARDED", "windowed_k512,recursive_full"),
                    help="comma list of workloads whose single evaluation is also column-sharded over the ranks (N>1 only); '' to skip")
    ap.add_argument("--chain-steps", type=int, default=int(os.environ.get("NHP_BENCH_CHAIN_STEPS", "50")),
                    help="mcmc! steps per rank of the config-5 leg (N>1 only); 0 to skip")
    ap.add_argument("--no-two-streams", dest="two_streams", action="store_false",
                    help="skip the leg with two independent evaluation streams sharing the GPU (N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-default-dispatch", action="store_true", help="skip the recursive=true legs (the reference's default call)")
    ap.add_argument("--no-batch", action="store_true", help="skip the batch / changing-parameter legs")
    args = ap.parse_args()

    import torch                       # first: libnhp.so must bind to torch's HIP runtime copy
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # NHP_BENCH_BACKEND=gloo lets the N>1 path be rehearsed on a one-GPU box (ranks share device 0)
    backend = os.environ.get("NHP_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local)
    tdev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as entry
    nhp = entry.load_package()
    ctx = nhp.Context(local)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    r = run_workload(nhp, ctx, args.workload, args.nodes, args.events, args.steps, args.warmup, sync)
    wall = torch.tensor([r["wall"]], dtype=torch.float64, device=tdev)
    lls = torch.tensor([r["ll"]], dtype=torch.float64, device=tdev)
    if world > 1:
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
        gathered = [torch.zeros_like(lls) for _ in range(world)] if rank == 0 else None
        dist.gather(lls, gathered, dst=0)                  # the per-chain results, over RCCL/xGMI
        if rank == 0:
            lls = torch.cat(gathered)
    wall_s = float(wall.item())
    out = headline(args, r, world, wall_s, lls) if rank == 0 else None

    # N > 1: the legs after the timed region (config 5's chains, one evaluation / one chain over all ranks) exchange through
    # RCCL communicators of the library's own.  They are secondary numbers and must never cost the headline line: an
    # exception is recorded in the line, and a leg that does not return (a rank stuck in a collective) is cut short by a
    # watchdog thread that prints the line as it stands and ends the process (ctypes and torch collectives release the GIL).
    watchdog = None
    state = {"printed": False, "partial": False}
    if world > 1:
        deadline = float(os.environ.get("NHP_BENCH_EXTRAS_DEADLINE_S", "300"))

        def give_up():
            # the line as it stands, marked partial, then a NON-ZERO exit: a leg that hangs must not read as a clean run
            if rank == 0 and not state["printed"]:
                out["multi_gpu_legs"] = {"error": f"not finished within {deadline:.0f} s: cut short, headline unaffected"}
                print(finish_line(out), flush=True)
            sys.stdout.flush()
            os._exit(3)
        watchdog = threading.Timer(deadline + (0.0 if rank == 0 else 10.0), give_up)
        watchdog.daemon = True
        watchdog.start()
    def rest():
        chain_wall = None
        if world > 1 and args.chain_steps > 0:
            # per rank: [own chain wall, one chain over all ranks, gather, library communicator world, its rank, raised]
            mine = torch.zeros(6, dtype=torch.float64, device=tdev)
            try:
                mine[:5] = torch.tensor(list(chains_leg(nhp, ctx, rank, args.chain_steps, sync)), dtype=torch.float64, device=tdev)
            except Exception as exc:                # recorded below; the other ranks see the flag
                mine[5] = 1.0
                print(f"[rank {rank}] config-5 leg failed: {exc!r}", file=sys.stderr, flush=True)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank = torch.stack(every).cpu()
            if float(per_rank[:, 5].max().item()) == 0.0:
                chain_wall, one_chain_wall, gather_wall = (float(per_rank[:, k].max().item()) for k in (0, 1, 2))
                gather_wall = -1.0 if float(per_rank[:, 2].min().item()) < 0.0 else gather_wall
                one_chain_wall = -1.0 if float(per_rank[:, 1].min().item()) < 0.0 else one_chain_wall
                chain_ranks = [{"torch_rank": q, "library_comm_world": int(per_rank[q, 3].item()), "library_comm_rank": int(per_rank[q, 4].item()),
                                "mcmc_steps_per_sec": args.chain_steps / float(per_rank[q, 0].item())} for q in range(world)]
            elif rank == 0:
                out["config5_independent_chains"] = {"error": "a rank raised in the chains leg (stderr has the message)"}
        sharded = None
        if world > 1 and args.sharded:
            try:
                sharded = sharded_leg(nhp, ctx, args.nodes, args.events, world, [x for x in args.sharded.split(",") if x])
            except Exception as exc:
                sharded = {"error": repr(exc)}
        if rank == 0:
            if chain_wall is not None:
                out["config5_independent_chains"] = {
                    "workload": "c3 model (N=1024, M=1e6, logit-normal network), one mcmc! chain per rank, device-side sweep",
                    "chains": world, "steps_per_chain": args.chain_steps,
                    # what the library's OWN communicator saw on every rank (world 0 = no RCCL clique: the gloo rehearsal)
                    "ranks": chain_ranks,
                    "rccl_clique_of_the_library": all(q["library_comm_world"] == world for q in chain_ranks)
                                                  and sorted(q["library_comm_rank"] for q in chain_ranks) == list(range(world)),
                    "mcmc_steps_per_sec": world * args.chain_steps / chain_wall, "ms_per_step": 1e3 * chain_wall / args.chain_steps,
                    "rccl_gather_of_the_chains_posterior_moments_ms": 1e3 * gather_wall if gather_wall >= 0.0 else None,
                    "one_chain_over_all_ranks_ms_per_step": 1e3 * one_chain_wall / args.chain_steps if one_chain_wall >= 0.0 else None}
                if gather_wall < 0.0:
                    out["config5_independent_chains"]["gather"] = {"error": "the gather of the chains' moments failed on a rank (stderr has the message)"}
                if one_chain_wall < 0.0:
                    out["config5_independent_chains"]["one_chain_over_all_ranks"] = {"error": "failed on a rank (stderr has the message)"}
            if sharded is not None:
                out["one_evaluation_over_all_ranks"] = sharded
            if world == 1 and args.two_streams and not r["recursive"]:
                try:
                    out["two_independent_streams_on_one_gpu"] = two_streams(nhp, ctx, r, args.steps, local)
                except Exception as exc:        # secondary number
                    out["two_independent_streams_on_one_gpu"] = {"error": repr(exc)}
            if world == 1 and not args.no_cpu:
                out["cpu_baseline"] = cpu_baseline(r)
                out["speedup_vs_cpu_core"] = out["value"] / out["cpu_baseline"]["value"]
                if not r["recursive"]:
                    try:
                        out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(r)
                        out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline_all_cores"]["value"]
                    except Exception as exc:      # secondary number
                        out["cpu_baseline_all_cores"] = {"error": repr(exc)}
            if world == 1 and not args.no_default_dispatch:
                out["default_dispatch"] = default_dispatch_leg(nhp, ctx, args, sync)
            if world == 1 and not args.no_batch:
                try:
                    out["batch"] = batch_leg(nhp, ctx, r, args, sync)
                except Exception as exc:        # secondary number: never take the headline down with it
                    out["batch"] = {"error": repr(exc)}
                try:
                    out["parameters_changing_every_evaluation"] = changing_parameters_leg(nhp, ctx, r, args, sync)
                except Exception as exc:
                    out["parameters_changing_every_evaluation"] = {"error": repr(exc)}
                try:
                    out["mle"] = mle_leg(nhp, ctx, args)
                except Exception as exc:
                    out["mle"] = {"error": repr(exc)}
            if world == 1 and args.extra:
                others = []
                for name in [s for s in args.extra.split(",") if s and s != args.workload]:
                    steps = max(3, args.steps // (10 if name in ("recursive", "recursive_full", "windowed_k512") else 2))
                    o = run_workload(nhp, ctx, name, args.nodes, args.events, steps, 2, sync)
                    Bo = algorithmic_bytes(o["N"], o["M"], o["kind"])
                    mk = o["dev_ms"] / steps
                    # exponential pair terms/s against the calibrated fp64-VALU ceiling (nhp_probe_rate, tools/rate.py);
                    # the recursive path evaluates 2·M·N exponentials per call
                    entry = {"workload": name, "events": o["M"], "value": steps / o["wall"], "kernel_ms": mk, "steps": steps,
                             "pairs_per_eval": o["pairs"], "hbm_frac": Bo / (mk * 1e-3) / 1e9 / HBM_PEAK_GBS, "loglik": o["ll"]}
                    if name != "recursive":        # (the default recursive path runs a model-dependent truncated window)
                        terms = 2.0 * o["M"] * o["N"] if name == "recursive_full" else float(o["pairs"])
                        entry["exp_terms_per_s"] = terms / (mk * 1e-3)
                        entry["fp64_valu_frac"] = terms / (mk * 1e-3) / EXP_TERM_CEILING
                    others.append(entry)
                out["other_workloads"] = others
            if world == 1 and args.configs:
                out["configs"] = config_workloads(nhp, ctx, args.configs.split(","))
            print(finish_line(out), flush=True)
            state["printed"] = True
            state["partial"] = out["status"] != "ok"
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()

    # past the headline nothing may cost the LINE: rank 0 prints what it has, marked "status": "partial" with the failed legs
    # named -- and then the process leaves NON-ZERO (3), so that a leg that raised or hung never reads as a clean run
    try:
        rest()
    except BaseException as exc:
        if world == 1:
            raise
        print(f"[rank {rank}] after the headline: {exc!r}", file=sys.stderr, flush=True)
        if rank == 0 and not state["printed"]:
            out["multi_gpu_legs"] = {"error": repr(exc)}
            print(finish_line(out), flush=True)
        sys.stdout.flush()
        os._exit(3)
    if watchdog is not None:
        watchdog.cancel()
    if world > 1 and state.get("partial"):          # a leg recorded an error inside the (printed) line
        sys.stdout.flush()
        os._exit(3)


if __name__ == "__main__":
    main()
