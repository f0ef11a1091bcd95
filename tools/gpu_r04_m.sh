#!/bin/bash
# Round-4 session M: the DC chain wave's issue priority on its own (SDRM_DC_PRIO), the clock stage's consumer staying at 3
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r04_dc_prio.txt"
export TMPDIR=/tmp
cd "$R"
: > $OUT
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1); printf "  %-34s %s\n" "$label" "$r" | tee -a "$OUT"; }
n=131072
echo "columns: ms per step, Msamples/s, front-end / DC / clock stage ms per launch (pipelined)" >> $OUT
for ch in 1024 2048 4096; do
  echo "== chunk $n x $ch channels" | tee -a "$OUT"
  for pr in 3 0 1 3 0 1; do
    cell "DC chain priority $pr (clock 3)" SDRM_DC_PRIO=$pr
  done
done
