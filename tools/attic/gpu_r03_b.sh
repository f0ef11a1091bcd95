#!/bin/bash
# round 3, call B: slim clock-stage shapes -- parity, then step times at many channels with and without the stream holds
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03b"
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
  for shape in 16 32 64 16x512 16x256 32x256; do
    run $ch $shape
  done
done
for ch in 2048 4096; do
  for shape in 64 16x256 32x256; do
    run $ch $shape SDRM_DC_FIRST=0
    run $ch $shape SDRM_FRONT_HOLD=128,8192
  done
done
