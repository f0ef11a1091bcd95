#!/bin/bash
# Round-4 session R: the calibration measuring >= 4 ms per candidate: the launch-bound cells of the grid again, then suite + bench
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r04_grid_short_calls.txt"
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
cd "$R"
: > $OUT
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 2>/dev/null | tail -1); printf "  %-40s %s\n" "$label" "$r" | tee -a "$OUT"; }
for n in 4096 16384; do
  for ch in 64 256 1024; do
    echo "== chunk $n x $ch channels" | tee -a "$OUT"
    for rep in 1 2 3; do
      cell "default (self-calibrated)" A=1
      cell "calibration off (channel-count rules)" SDRM_AUTOTUNE=0
    done
  done
done
timeout 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method=thread > gpurun_out/r04_pytest_r.log 2>&1; echo "suite exit $?"; grep -E "passed|failed" gpurun_out/r04_pytest_r.log | tail -1
timeout 600 python bench.py > gpurun_out/r04_bench_r.json 2> gpurun_out/r04_bench_r.err; echo "bench exit $?"; cut -c1-300 gpurun_out/r04_bench_r.json
