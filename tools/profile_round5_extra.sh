#!/bin/bash
# Runs ON THE GPU BOX (via gpurun), after tools/profile_round.sh: the persistent few-evaluation launch under rocprofv3 (kernel durations of
# single objective(alpha, rho) calls at N = 1024 and N = 4096), and the default bench line with its cpu_baseline leg.
# usage: bash tools/profile_round5_extra.sh <tag>      -> gpurun_out/<tag>/...
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/chain_stats -- python3 tools/latency.py --sizes 512,2048 --batches 1,2 --reps 40 > $out/chain_stats.log 2>&1 || exit 1
timeout -k 10 400 python3 bench.py > $out/bench_default.log 2>&1 || exit 1
grep "^{\"metric\"" $out/bench_default.log > $out/bench_line_default.json
