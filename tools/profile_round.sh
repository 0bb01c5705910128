#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench command, then
# separate PMC passes (never combined with other trace domains) on one 256-evaluation group.
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/stats.log 2>&1 || exit 1
grep "^{\"metric\"" $out/stats.log > $out/bench_line.json
for pm in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $pm | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $out/pmc_$t -- python3 bench.py --steps 1 --warmup 0 --grid 256 --no-cpu-baseline --no-roofline > $out/pmc_$t.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py $out > $out/summary.json && cat $out/summary.json
