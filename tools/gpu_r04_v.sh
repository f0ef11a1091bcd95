#!/bin/bash
# Round-4 session V: the online refinement for calls with NCO batches: configs[4]'s mix default / calibration off, the bench-shape tests
set +e
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
mkdir -p gpurun_out
O=gpurun_out/r04_config5_online.txt
: > $O
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/config5.py $ch 2>&1 | grep -E "channels:|refined" | sed 's/.*channels: //' | tr '\n' '|'); printf "  %4d ch  %-28s %s\n" $ch "$label" "$r" | tee -a $O; }
for ch in 256 512 1024; do
  for rep in 1 2 3; do
    cell "default (refined online)" SDRM_AUTOTUNE_LOG=1
    cell "calibration off" SDRM_AUTOTUNE=0
  done
done
timeout 900 python -u -m pytest tests/test_gpu_bench_shapes.py -m gpu -x -q --timeout 250 --timeout-method=thread 2>&1 | tail -3
timeout 900 python -u -m pytest tests -m gpu -x -q --timeout 250 --timeout-method=thread > gpurun_out/r04_pytest_v.log 2>&1; echo "suite exit $?"; grep -E "passed|failed" gpurun_out/r04_pytest_v.log | tail -1
timeout 600 python bench.py > gpurun_out/r04_bench_v.json 2> gpurun_out/r04_bench_v.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_bench_v.json') if l.startswith('{')][-1])
print(d['value'], d['verified_vs_oracle'], d['config5']['value'], d['config5']['verified_vs_oracle'], d['config5'].get('schedule'))
PY
