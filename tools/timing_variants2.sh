#!/bin/bash
# The FIXED cost of a gpcc_update_solve job (41.5 us of prologue + epilogue per job beside ~29 us per tile product): TIMING-ONLY
# builds (wrong results, tools/_timing_libs/, never shipped) that each drop one part -- the accumulator initialisation, the DMA of inv(L_kk),
# the whole panel solve, the output staging + store -- run back to back with the real library on one box, at N = 4096 and N = 2048.
#  for v in NO_XLOAD NO_SOLVE NO_STORE NO_INIT; do GPCC_HIP_LIB=tools/_timing_libs/libgpcc_$v.so GPCC_BUILD_DEFINES="-DGPCC_TIMING_$v" python3 -c "from gpcc_amd import build; build.build(force=True)"; done
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for tag in real NO_INIT NO_XLOAD NO_SOLVE NO_STORE; do
    lib=$GRAFT_REPO_ROOT/tools/_timing_libs/libgpcc_$tag.so; [ $tag = real ] && lib=""
    defs=""; [ $tag != real ] && defs="-DGPCC_TIMING_$tag"
    for args in "--steps 3" "--steps 10 --n-per-band 1024 --grid 256"; do
      GPCC_BUILD_DEFINES="$defs" GPCC_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', '$args', '|', d['value'], 'evals/s', d['ms_per_step'], 'ms/step', r['kernels_ms'], 'info_nonzero', d['info_nonzero'])"
    done
  done
done
