#!/bin/bash
# BASELINE.json configs on one GPU (per-GPU share of the grids of the 8-GPU configs)
run() { echo "== $1"; shift; timeout -k 10 500 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"], "|", d["value"], "evals/s |", d["ms_per_step"], "ms/step | update", r["achieved"], "TF", r["frac"])'; }
run "cfg2  2x1024 matern32 fp64 G=256"        --n-per-band 1024 --grid 256 --steps 5
for k in OU rbf matern32 matern52; do run "cfg3  2x2048 $k fp64 G=1024" --kernel $k --steps 2; done
run "cfg4  3x1365 matern32 fp64 G=8192 (1/8 of 256x256)" --bands 3 --n-per-band 1365 --grid 8192 --steps 1 --warmup 1
run "cfg5  2x8192 matern52 fp32 G=512 (1/8 of 4096)" --precision fp32 --n-per-band 8192 --kernel matern52 --grid 512 --steps 1 --warmup 1
