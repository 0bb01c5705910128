#!/bin/bash
# A/B on ONE box of the two forms of an LDS-DMA piece: the scalar-address inline-asm form (default) against the compiler's builtin
# with per-lane pointers (-DGPCC_AB_DMA_BUILTIN, rounds 1-3).  The second library is built in the build container:
#   GPCC_HIP_LIB=$PWD/tools/_ab_libs/libgpcc_dma_builtin.so GPCC_BUILD_DEFINES=-DGPCC_AB_DMA_BUILTIN python3 -c "from gpcc_amd import build; build.build(force=True)"
# (tools/_ab_libs/ is git-ignored and travels with the gpurun snapshot).  Runs ON THE GPU BOX.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for tag in scalar builtin; do
    lib=""; [ $tag = builtin ] && lib=$GRAFT_REPO_ROOT/tools/_ab_libs/libgpcc_dma_builtin.so
    for args in "--steps 4" "--steps 4 --precision fp32" "--steps 10 --n-per-band 1024 --grid 256"; do
      defs=""; [ $tag = builtin ] && defs="-DGPCC_AB_DMA_BUILTIN"    # (should bench.py find the library stale and rebuild it: with its define)
      GPCC_BUILD_DEFINES=$defs GPCC_HIP_LIB=$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', '$args', '|', d['value'], 'evals/s |', d['ms_per_step'], 'ms/step |', r['kernels_ms'])"
    done
  done
done
