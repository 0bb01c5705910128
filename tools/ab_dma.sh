#!/bin/bash
# A/B on ONE box of the operand-staging variants of the MFMA kernels (round 4):
#   default  LDS-DMA pieces in scalar-address inline-asm form
#   builtin  -DGPCC_AB_DMA_BUILTIN: the compiler's builtin with per-lane pointers (rounds 1-3)
# The variant libraries are built in the build container into tools/_ab_libs/ (git-ignored; travels with the gpurun snapshot):
#   GPCC_HIP_LIB=$PWD/tools/_ab_libs/libgpcc_<tag>.so GPCC_BUILD_DEFINES="<defines>" python3 -c "from gpcc_amd import build; build.build(force=True)"
# Runs ON THE GPU BOX.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for tag in default builtin; do
    lib=""; defs=""
    [ $tag = builtin ] && lib=$GRAFT_REPO_ROOT/tools/_ab_libs/libgpcc_builtin.so && defs="-DGPCC_AB_DMA_BUILTIN"
    for args in "--steps 4" "--steps 4 --precision fp32" "--steps 10 --n-per-band 1024 --grid 256"; do
      GPCC_BUILD_DEFINES=$defs GPCC_HIP_LIB=$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', '$args', '|', d['value'], 'evals/s |', d['ms_per_step'], 'ms/step |', r['kernels_ms'], d['info_nonzero'])"
    done
  done
done
