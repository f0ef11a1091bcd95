#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
cd $R
mkdir -p gpurun_out
O=gpurun_out/r04_tune_diag.txt
: > $O
for ch in 256 1024; do for n in 4096 16384; do
  echo "== $n x $ch, every block the starting point" | tee -a $O
  SDRM_TUNE_DIAG=1 SDRM_AUTOTUNE_LOG=1 timeout 300 python tools/sweep_cell.py $ch $n 200 131072 2>&1 | grep -E "^[0-9]|refined|calibrated" | tee -a $O
done; done
