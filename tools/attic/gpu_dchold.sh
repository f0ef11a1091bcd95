#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03_dchold.txt"
export TMPDIR=/tmp
cd "$R"
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1); printf "  %-34s %s\n" "$label" "$r" | tee -a "$OUT"; }
for n in 32768 131072; do
  for ch in 1536 2048 3072 4096; do
    echo "== chunk $n x $ch channels" | tee -a "$OUT"
    cell "default (dc waits for clock >= 2048)" A=1
    cell "dc never waits" SDRM_DC_HOLD=0
    cell "default (again)" A=1
    cell "dc never waits (again)" SDRM_DC_HOLD=0
    cell "dc waits from 1024" SDRM_DC_HOLD=1024
  done
done
