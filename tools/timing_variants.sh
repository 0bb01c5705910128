#!/bin/bash
# Where does the update kernel's loop lose its 6 %?  TIMING-ONLY builds of libgpcc_hip.so (wrong results, built into /tmp,
# never shipped) that each remove ONE ingredient of gpcc_update_solve's main loop, run back to back with the real library on one box.
# Runs ON THE GPU BOX via gpurun.   usage: bash tools/timing_variants.sh > gpurun_out/timing_variants.log
cd $GRAFT_REPO_ROOT
# (the variant libraries are built in the build container into tools/_timing_libs/, git-ignored, and travel with the snapshot:
#  for v in ...; do GPCC_HIP_LIB=tools/_timing_libs/libgpcc_<tag>.so GPCC_BUILD_DEFINES="-DGPCC_TIMING_<v>" python3 -c "from gpcc_amd import build; build.build(force=True)"; done)
for rep in 1 2; do
  for tag in real NO_BARRIER NO_DMA NO_DMA+NO_BARRIER; do
    lib=$GRAFT_REPO_ROOT/tools/_timing_libs/libgpcc_$tag.so; [ $tag = real ] && lib=""
    GPCC_BUILD_DEFINES="$(echo $tag | sed 's/^real$//; s/NO_/-DGPCC_TIMING_NO_/g; s/+/ /g')" GPCC_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], 'evals/s', d['ms_per_step'], 'ms/step', r['kernels_ms'], 'info_nonzero', d['info_nonzero'], d['clock_mhz'], d['power_w'])"
  done
done
