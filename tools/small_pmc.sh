#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 PMC passes (never combined with other trace domains) of the small-N kernel family at
# the reference's documented sizes, condensed by tools/small_pmc_summary.py.   usage: bash tools/small_pmc.sh <tag>
tag=${1:-small_pmc}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for cfg in "--n-per-band 55 --grid 12321" "--n-per-band 50 --bands 3 --grid 16384" "--n-per-band 55 --grid 400"; do
  i=$((i+1))
  p=0
  for pm in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_LEVEL_WAVES"; do
    p=$((p+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pm --output-format csv -d $out/cfg${i}_pass$p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $cfg > $out/cfg${i}_pass$p.log 2>&1 || exit 1
  done
done
python3 tools/small_pmc_summary.py $out > $out/summary.json && cat $out/summary.json
