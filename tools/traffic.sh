#!/bin/bash
# HBM-side traffic of the dominant kernels from rocprofv3 PMC passes (separate passes, --kernel-trace only), written to
# gpurun_out/traffic.json in the format bench.py reads from profiles/traffic.json.  Run on the GPU box:
#   NHP_HEAD=$(git rev-parse --short HEAD) gpurun -- "NHP_HEAD=$NHP_HEAD bash tools/traffic.sh"
# then copy gpurun_out/traffic.json (and the raw counter CSVs under gpurun_out/pmc_traffic/) into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {   # run <tag> <counters...> -- <cmd...>
  tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $R/gpurun_out/pmc_traffic/$tag -- "$@" > /dev/null 2>&1
}
# calibration of FETCH_SIZE on the child slices' own access shape (4 + 2 bytes per lane; the guide's factor 2 is for 16 bytes per
# lane): nhp_probe_stream mode 1 reads a known number of bytes per launch
STREAM_ONE=1,512,1024,512 run cal_fetch FETCH_SIZE -- python3 $R/tools/dbg/streambw.py
STREAM_ONE=1,512,1024,512 python3 $R/tools/dbg/streambw.py > $R/gpurun_out/pmc_traffic/cal_bytes.txt 2>/dev/null
run k8_fetch FETCH_SIZE -- python3 $R/tools/kbench.py windowed_k8 10
run k8_write WRITE_SIZE -- python3 $R/tools/kbench.py windowed_k8 10
run k8_tcc TCC_HIT_sum TCC_MISS_sum -- python3 $R/tools/kbench.py windowed_k8 10
NB=8 run b8_fetch FETCH_SIZE -- python3 $R/tools/batch.py
NB=8 run b8_write WRITE_SIZE -- python3 $R/tools/batch.py
NB=8 run b8_tcc TCC_HIT_sum TCC_MISS_sum -- python3 $R/tools/batch.py
python3 - <<PY
import csv, glob, json, collections, os
R = "$R"
def mean(tag, kern, name):
    v = [float(r["Counter_Value"]) for f in glob.glob(f"{R}/gpurun_out/pmc_traffic/{tag}/*/*counter_collection.csv")
         for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v) if v else None
out = {}
# bytes per FETCH_SIZE KB on the slices' access shape: the probe's known bytes per launch over its counter value
cal = None
try:
    known = int(open(f"{R}/gpurun_out/pmc_traffic/cal_bytes.txt").read().split("bytes=")[1].split()[0])
    cf = mean("cal_fetch", "k_probe_stream", "FETCH_SIZE")
    cal = known / (cf * 1024.0)
except Exception as exc:
    print("calibration pass failed:", exc)
for key, pre, kern, label, factor in (("windowed_k8", "k8", "k_windowed_slices<", "k_windowed_slices<512,2,true,false>", cal),
                                      ("batch_4_sets", "b8", "k_slices_batch<", "k_slices_batch<512,2,4>", cal)):
    f, w = mean(pre + "_fetch", kern, "FETCH_SIZE"), mean(pre + "_write", kern, "WRITE_SIZE")
    h, m = mean(pre + "_tcc", kern, "TCC_HIT_sum"), mean(pre + "_tcc", kern, "TCC_MISS_sum")
    if f is None or factor is None: continue
    out[key] = {"kernel": label, "commit": os.environ.get("NHP_HEAD", "unknown"),
                "source": f"gpurun_out/pmc_traffic/{pre}_{{fetch,write,tcc}} (separate rocprofv3 --pmc passes; tools/traffic.sh)",
                "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "fetch_factor": factor,
                "correction": ("FETCH_SIZE calibrated on this kernel's own access shape (nhp_probe_stream mode 1: 4 + 2 bytes per lane, known bytes per launch; "
                               "MI355X_MICROARCH.md HBM: other widths than 16 B per lane are uncalibrated): factor = known bytes / counter"
                               if factor is cal else
                               "MI355X_MICROARCH.md HBM: FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> doubled") + "; WRITE_SIZE exact; unit KB",
                "traffic_bytes_per_launch": int(factor * f * 1024 + (w or 0) * 1024), "tcc_hit_rate": h / (h + m) if h is not None else None}
json.dump(out, open(f"{R}/gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
