#!/bin/bash
# Same-box A/B of small-N kernel variants: bash tools/ab_small.sh "<tag>:<defines>" ... (tag "default" = the shipped library; the others are
# tools/_ab_libs/libgpcc_<tag>.so, built in the build container with GPCC_BUILD_DEFINES="<defines>").  Runs ON THE GPU BOX.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in "$@"; do
    tag=${v%%:*}; defs=${v#*:}; [ "$defs" = "$v" ] && defs=""
    lib=""; [ $tag != default ] && lib=$GRAFT_REPO_ROOT/tools/_ab_libs/libgpcc_$tag.so
    for args in "--n-per-band 55 --grid 12321 --steps 20" "--n-per-band 75 --grid 12321 --steps 20" "--n-per-band 50 --bands 3 --grid 12321 --steps 20" "--n-per-band 65 --grid 12321 --steps 20"; do
      GPCC_BUILD_DEFINES=$defs GPCC_HIP_LIB=$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$args', '|', d['value'], 'evals/s |', d['ms_per_step'], 'ms/step |', d['info_nonzero'])"
    done
    GPCC_BUILD_DEFINES=$defs GPCC_HIP_LIB=$lib timeout -k 10 300 python3 tools/readme_bench.py --sweeps A,C 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag sweep', d['sweep'], d.get('fit_seconds'), 's fit |', d['fixed_hyper_ms_per_batch'], 'ms/batch')"
  done
done
