"""Config 3 (N=1024, M=1e6, logit-normal network): the resident mcmc! chain (nhp_cont_mcmc_run), steps per call from argv.
Run under rocprofv3 --kernel-trace --stats for the per-kernel breakdown of a step."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nhp = e.load_package()
from nhp_amd import _lib, inference
ctx = nhp.Context(0)
N, M = 1024, 1_000_000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
proc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
ds = nhp.device_dataset(proc, (times, nodes, T), ctx)
model, pri = proc.device_model(ctx), inference._priors(proc)
_lib.check(_lib.lib().nhp_cont_model_set_rho(ctx.h, model.h, 0.5), ctx.h)
s0 = 0
for rep in range(4):
    t0 = time.perf_counter()
    _lib.check(_lib.lib().nhp_cont_mcmc_run(ctx.h, None, ds.h, model.h, C.byref(pri), 1.0, 1.0, 1, s0, steps, 0), ctx.h)
    s0 += steps
    print(f"{steps} steps: {1e3 * (time.perf_counter() - t0) / steps:.4f} ms per mcmc! step")
