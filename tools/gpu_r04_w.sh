#!/bin/bash
# Round-4 session W: online refinement by call class: short calls on a long-buffer batch, configs[4]'s mix (default and forced clock-stage
# shapes), tests, bench
set +e
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
R=${GRAFT_REPO_ROOT:?}
cd $R
mkdir -p gpurun_out
O=gpurun_out/r04_online_short_calls.txt
: > $O
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch $n 200 131072 2>&1 | grep -E "^[0-9]|refined" | sed 's/sdrmodem_hip: //' | tr '\n' '|'); printf "  %5d x %4d ch  %-26s %s\n" $n $ch "$label" "$r" | tee -a $O; }
for n in 4096 16384 32768; do
  for ch in 256 1024; do
    for rep in 1 2; do
      cell "default (refined online)" SDRM_AUTOTUNE_LOG=1
      cell "companion grid forced off" SDRM_K3_COMPANY=0,0,0
      cell "companion grid forced on" SDRM_K3_COMPANY=4096,1,100000
    done
  done
done
O2=gpurun_out/r04_config5_shapes.txt
: > $O2
for rep in 1 2; do
for sh in default 16x512 16x256 32x512; do
  if [ $sh = default ]; then e="A=1"; else e="SDRM_K3_LANES=$sh"; fi
  r=$(env $e SDRM_AUTOTUNE_LOG=1 timeout 300 python tools/config5.py 256 2>&1 | grep -E "channels:|refined" | sed 's/.*channels: //; s/sdrmodem_hip: refined online for calls with NCO batches, [0-9]* samples per call: //' | tr '\n' '|')
  printf "  %-8s %s\n" $sh "$r" | tee -a $O2
done
done
timeout 900 python -u -m pytest tests -m gpu -x -q --timeout 250 --timeout-method=thread > gpurun_out/r04_pytest_w.log 2>&1; echo "suite exit $?"; grep -E "passed|failed" gpurun_out/r04_pytest_w.log | tail -1
timeout 600 python bench.py > gpurun_out/r04_bench_w.json 2> gpurun_out/r04_bench_w.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_bench_w.json') if l.startswith('{')][-1])
print(d['value'], d['verified_vs_oracle'], d['config5']['value'], d['config5']['verified_vs_oracle'], d['config5'].get('schedule'))
PY
