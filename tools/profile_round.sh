#!/bin/bash
# The round's profile set in one call: gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02c'
# Writes gpurun_out/<tag>_*.{json,csv}; copy what is to be judged into profiles/.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <name> <script and args...>: kernel stats csv -> $O/${TAG}_<name>_kernel_stats.csv
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$n -- python3 "$@" > $O/${TAG}_${n}_profiled.out 2> $O/${TAG}_${n}_profiled.err
  cp $O/prof_${TAG}_$n/*/*kernel_stats.csv $O/${TAG}_${n}_kernel_stats.csv
}
prof bench $R/bench.py
prof headline_only $R/bench.py --no-batch --no-default-dispatch --no-two-streams --extra '' --configs '' --no-cpu
prof config3_chain $R/tools/chain.py 50
prof config4 $R/tools/c4.py
DG_RATE=0.05 prof discrete_gibbs_parents $R/tools/dgibbs.py
cd $R && NHP_HEAD=${NHP_HEAD:-unknown} bash tools/traffic.sh > /dev/null 2>&1
ls $O | grep "^${TAG}_" | head -30
