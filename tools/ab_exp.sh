#!/bin/bash
# A/B of the two exp implementations on ONE box: builds a second library with -DGPCC_AB_POLY_EXP (degree-13 polynomial exp, rounds 1-2)
# beside the default one (2^(j/64) table + degree-5 polynomial) and runs the same bench lines with each.  Run via gpurun.
set -e
cd $GRAFT_REPO_ROOT
# the A/B library through the package's own build (ten objects, gpcc.jl_amd/build.py), into its own file
GPCC_HIP_LIB=/tmp/libgpcc_poly.so GPCC_BUILD_DEFINES="-DGPCC_AB_POLY_EXP" python3 -c "import sys; sys.path.insert(0, '.'); from gpcc_amd import build; print(build.build(force=True))"
for rep in 1 2; do
for lib in "" /tmp/libgpcc_poly.so; do
  tag=${lib:+poly}; tag=${tag:-table}
  for args in "--steps 3" "--precision fp32 --steps 3" "--n-per-band 55 --grid 12321 --steps 20" "--n-per-band 75 --bands 2 --grid 12321 --steps 20"; do
    GPCC_HIP_LIB=$lib python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', '$args', '|', d['value'], 'evals/s |', d['ms_per_step'], 'ms/step |', r['kernels_ms'])"
  done
done
done
