#!/bin/bash
# Generic same-box A/B of library variants: bash tools/ab_libs.sh "<tag>:<defines>" ... (tag "default" = the shipped library; the variant
# libraries tools/_ab_libs/libgpcc_<tag>.so are built in the build container with GPCC_BUILD_DEFINES="<defines>").  Runs ON THE GPU BOX.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "$@"; do
    tag=${v%%:*}; defs=${v#*:}; [ "$defs" = "$v" ] && defs=""
    lib=""; [ $tag != default ] && lib=$GRAFT_REPO_ROOT/tools/_ab_libs/libgpcc_$tag.so
    for args in "--steps 4" "--steps 4 --precision fp32" "--steps 10 --n-per-band 1024 --grid 256"; do
      GPCC_BUILD_DEFINES=$defs GPCC_HIP_LIB=$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', '$args', '|', d['value'], 'evals/s |', d['ms_per_step'], 'ms/step |', r['kernels_ms'], d['info_nonzero'])"
    done
  done
done
