#!/bin/bash
# One-GPU rehearsal of the N>1 bench path: 2 ranks share GPU 0.  First with RCCL (may refuse duplicate
# devices), then with gloo for the collective.  Small grid so that both workspaces fit.
export GPCC_BENCH_OVERSUBSCRIBE=1
for be in nccl gloo; do
  echo "== backend $be"
  GPCC_BENCH_BACKEND=$be timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --grid 128 --slots 64 --no-cpu-baseline --no-roofline 2>&1 | tail -4
done
