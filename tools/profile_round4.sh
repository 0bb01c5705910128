#!/bin/bash
# Round-4 profile set, runs ON THE GPU BOX: headline kernel stats + PMC (tools/profile_round.sh), the same counters for the one-launch
# step (option step_fused = 1), the fp32 headline's kernel stats, every BASELINE.json configuration on one GPU.
bash tools/profile_round.sh r4_prof > gpurun_out/r4_prof.log 2>&1
out=gpurun_out/r4_prof_step; mkdir -p $out; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pm in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $pm | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $out/pmc_$t -- python3 bench.py --steps 1 --warmup 0 --grid 256 --no-cpu-baseline --no-roofline --option step_fused=1 > $out/pmc_$t.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py $out > $out/summary.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_prof_fp32 -- python3 bench.py --no-cpu-baseline --precision fp32 > gpurun_out/r4_prof_fp32.log 2>&1
bash tools/bench_configs.sh > gpurun_out/r4_bench_configs.log 2>&1
