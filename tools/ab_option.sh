#!/bin/bash
# Same-box A/B of ONE handle option on bench lines: bash tools/ab_option.sh <key> <v1> <v2> [reps] [extra bench args ...].  Runs ON THE GPU BOX.
cd $GRAFT_REPO_ROOT
key=$1; v1=$2; v2=$3; reps=${4:-3}; shift 4
extra="$*"
for rep in $(seq 1 $reps); do
  for v in $v1 $v2; do
    for args in "--steps 4" "--steps 10 --n-per-band 1024 --grid 256" "--steps 4 --kernel OU" "--steps 4 --kernel rbf"; do
      timeout -k 10 400 python3 bench.py --no-cpu-baseline --option $key=$v $args $extra 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$key=$v', '$args $extra', '|', d['value'], 'evals/s |', d['ms_per_step'], 'ms/step |', r['kernels_ms'], d['info_nonzero'])"
    done
  done
done
