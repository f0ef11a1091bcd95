#!/bin/bash
# round 3, call A: the two-rank tests (gloo ranks sharing cuda:0 on a one-GPU box), baseline sweep points of the day
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03a"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
SDRM_KEEP_BENCH_LINE="$OUT/bench_gloo2.json" timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "two_ranks or bench_with_two" > "$OUT/pytest_two_ranks.log" 2>&1; echo "pytest exit $?"; tail -15 "$OUT/pytest_two_ranks.log"
for ch in 256 1024 2048 4096; do
  timeout 300 python tools/sweep_point.py $ch > "$OUT/sweep_$ch.txt" 2>&1; echo "sweep $ch exit $?"; head -1 "$OUT/sweep_$ch.txt"
done
