#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03_chain_prio.txt"
export TMPDIR=/tmp
cd "$R"
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1); printf "  %-34s %s\n" "$label" "$r" | tee -a "$OUT"; }
n=131072
for ch in 256 1024 2048 4096; do
  echo "== chunk $n x $ch channels" | tee -a "$OUT"
  for pr in 3 2 1 0 3 1; do
    cell "chain priority $pr" SDRM_CHAIN_PRIO=$pr
  done
done
