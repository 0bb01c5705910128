#!/bin/bash
# Same-box A/B of the shipped library against tools/_ab_libs/libgpcc_base.so (built from the previous commit: git stash; GPCC_HIP_LIB=...
# python -c "from gpcc_amd import build; build.build(force=True)"; git stash pop; touch both): bash tools/ab_base.sh [reps] "<bench args>" ...
cd $GRAFT_REPO_ROOT
reps=$1; shift
for rep in $(seq 1 $reps); do
  for tag in default base; do
    lib=""; [ $tag != default ] && lib=$GRAFT_REPO_ROOT/tools/_ab_libs/libgpcc_$tag.so
    for args in "$@"; do
      GPCC_HIP_LIB=$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', '$args', '|', d['value'], 'evals/s |', d['ms_per_step'], 'ms/step |', r['kernels_ms'], d['info_nonzero'])"
    done
  done
done
