#!/bin/bash
# round 3, call C: the front-end's short discriminator form -- probe + parity subset, then kernel times
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03c"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "probe or fixtures or lucky7 or ragged or nan_inf or mixed_rate or many_channel or perf_configuration" > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"; tail -5 "$OUT/pytest.log"
timeout 300 python tools/stage_times.py 256 > "$OUT/stage_256.txt" 2>&1; tail -4 "$OUT/stage_256.txt"
for ch in 256 4096; do
  timeout 300 python tools/sweep_point.py $ch 2>/dev/null | head -1 | tee -a "$OUT/sweep.txt"
done
