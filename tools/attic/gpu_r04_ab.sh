#!/bin/bash
# Round-4 session AB: issue priority of the clock stage's staging (producer) wave: 0 (as it was) / 1 / 2
# columns: ms per step, Msamples/s, front-end / DC / clock stage ms per launch
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:?}
mkdir -p gpurun_out
O=gpurun_out/r04_k3_producer_prio.txt
: > $O
for ch in 4096 3072 2048 1024 256; do
  for pr in 0 2 1 0 2; do
    r=$(SDRM_K3_PRODUCER_PRIO=$pr timeout 300 python tools/sweep_cell.py $ch 131072 2>/dev/null | tail -1)
    printf "  %5d ch  staging wave priority %d   %s\n" $ch $pr "$r" | tee -a $O
  done
done
