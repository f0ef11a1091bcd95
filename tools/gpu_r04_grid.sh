#!/bin/bash
# Round 4: the self-calibrated default of every cell of the heuristics grid (chunk x channels) next to every forced setting of
# the dimensions the calibration decides (review item 5: no cell more than 3 % behind its best forced setting).
# columns: ms per step, Msamples/s, front-end / DC / clock stage ms per launch
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r04_heuristics_grid.txt"
export TMPDIR=/tmp
cd "$R"
: > $OUT
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1); printf "  %-40s %s\n" "$label" "$r" | tee -a "$OUT"; }
for n in 4096 32768 131072 262144; do
  for ch in 64 256 1024 4096; do
    echo "== chunk $n x $ch channels" | tee -a "$OUT"
    cell "default (self-calibrated)" A=1
    cell "default (again)" A=1
    cell "calibration off (channel-count rules)" SDRM_AUTOTUNE=0
    cell "front hold forced on" SDRM_FRONT_HOLD=1,100000
    cell "front hold forced off" SDRM_FRONT_HOLD=0,0
    if [ $ch -le 2048 ]; then
      cell "companion grid forced on" SDRM_K3_COMPANY=4096,1,100000
      cell "companion grid forced off" SDRM_K3_COMPANY=0,0,0
    fi
    if [ $ch -ge 512 ]; then
      cell "clock stage 16x1024" SDRM_K3_LANES=16x1024
      cell "clock stage 32x512" SDRM_K3_LANES=32x512
      cell "clock stage 64x256p" SDRM_K3_LANES=64x256p
    fi
  done
done
