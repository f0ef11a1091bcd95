#!/bin/bash
# The scheduling switches off their tuning point: chunk x channels x {default, each switch forced off / to its alternative}.
# ms per step (Msamples/s) [front, dc, clock kernel ms]; one process per cell and variant.
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/${1:-r03_heuristics.txt}"
CHUNKS=${2:-"4096 32768 131072 262144"}
CHANS=${3:-"64 256 1024 4096"}
export TMPDIR=/tmp
cd "$R"
cell() { # label env...
  label=$1; shift
  r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1)
  printf "  %-34s %s\n" "$label" "$r" | tee -a "$OUT"
}
for n in $CHUNKS; do
  for ch in $CHANS; do
    echo "== chunk $n x $ch channels" | tee -a "$OUT"
    cell "default" A=1
    cell "no front hold" SDRM_FRONT_HOLD=0,0
    cell "front hold always" SDRM_FRONT_HOLD=1,100000
    cell "no dc-first hold" SDRM_DC_FIRST=0
    cell "dc-first always" SDRM_DC_FIRST=1,60
    cell "no companion grid" SDRM_K3_COMPANY=0,0,0
    cell "companion 64 always" SDRM_K3_COMPANY=64,1,100000
    cell "companion 4096 always" SDRM_K3_COMPANY=4096,1,100000
    cell "lanes 16" SDRM_K3_LANES=16
    cell "lanes 32" SDRM_K3_LANES=32
    cell "lanes 64" SDRM_K3_LANES=64
    cell "lanes 64x256p" SDRM_K3_LANES=64x256p
  done
done
