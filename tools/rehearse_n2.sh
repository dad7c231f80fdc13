#!/bin/bash
# The N>1 bench path rehearsed on a ONE-GPU box: two ranks over gloo sharing device 0 (no RCCL clique can form: the library's
# communicator reports world 0).  Writes the clean line and a line cut short by the watchdog, each with its exit status:
#   gpurun -- 'bash tools/rehearse_n2.sh r03_rehearsal'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-rehearsal}
mkdir -p $O
cd $R
run() {  # run <name> <deadline> <chain steps>
  NHP_BENCH_BACKEND=gloo NHP_BENCH_EXTRAS_DEADLINE_S=$2 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 --chain-steps $3 > $O/$1.out 2> $O/$1.err
  echo "exit status $?" >> $O/$1.out
  tail -c 1500 $O/$1.out
}
run clean 600 5
run cut_short 0.05 50
