#!/bin/bash
# second pass of the heuristics sweep: the default of every cell, three times (run-to-run spread), next to the switch that
# won the first pass
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/${1:-r03_heuristics_after.txt}"
export TMPDIR=/tmp
cd "$R"
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1); printf "  %-34s %s\n" "$label" "$r" | tee -a "$OUT"; }
for n in 4096 32768 131072 262144; do
  for ch in 64 256 1024 4096; do
    echo "== chunk $n x $ch channels" | tee -a "$OUT"
    cell "default" A=1
    cell "default (again)" A=1
    cell "default (third)" A=1
    cell "no front hold" SDRM_FRONT_HOLD=0,0
    cell "no dc-first hold" SDRM_DC_FIRST=0
    cell "no companion grid" SDRM_K3_COMPANY=0,0,0
  done
done
