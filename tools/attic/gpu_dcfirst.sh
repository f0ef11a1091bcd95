#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03_dcfirst.txt"
export TMPDIR=/tmp
cd "$R"
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1); printf "  %-34s %s\n" "$label" "$r" | tee -a "$OUT"; }
for n in 32768 131072; do
  for ch in 1536 2048 2560 3072 4096; do
    echo "== chunk $n x $ch channels" | tee -a "$OUT"
    cell "default" A=1
    cell "no dc-first hold" SDRM_DC_FIRST=0
    cell "default (again)" A=1
    cell "no dc-first hold (again)" SDRM_DC_FIRST=0
    cell "dc-first, 100 us bound" SDRM_DC_FIRST=1536,100
  done
done
