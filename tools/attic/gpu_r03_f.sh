#!/bin/bash
# round 3, call F: plain-ring clock stage (64 channels x 256 samples in 75 KB): parity, then step times
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03f"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "workgroup_shapes" > "$OUT/pytest_shapes.log" 2>&1; echo "pytest exit $?"; tail -3 "$OUT/pytest_shapes.log"
run() { # channels shape extra-env...
  ch=$1; shape=$2; shift 2
  line=$(env SDRM_K3_LANES=$shape "$@" timeout 200 python tools/sweep_point.py $ch 2>/dev/null | head -1)
  echo "$ch $shape $* : $line" | tee -a "$OUT/shapes.txt"
}
for ch in 1024 1536 2048 3072 4096; do
  for shape in 32 64 64x256p 32x256p; do
    run $ch $shape
  done
done
for ch in 2048 4096; do
  for shape in 64x256p; do
    run $ch $shape SDRM_DC_FIRST=0
    run $ch $shape SDRM_DC_FIRST=1536,30
    run $ch $shape SDRM_DC_FIRST=1536,100
  done
done
SDRM_K3_LANES=64x256p timeout 200 python tools/sweep_point.py 4096 > "$OUT/sweep_4096_plain.txt" 2>&1
