#!/bin/bash
# Round-4 session N: the GPU suite six times over (a sporadic hang seen once in session B never showed again: the watchdog
# would name the test), then a full-length soak with new seeds.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
: > gpurun_out/r04_suite_repeat.txt
for i in 1 2 3 4 5 6; do
  timeout 900 python -u -m pytest tests -m gpu -q --timeout 200 --timeout-method=thread -p no:cacheprovider > gpurun_out/r04_pytest_n$i.log 2>&1
  echo "run $i: exit $? $(grep -E 'passed|failed' gpurun_out/r04_pytest_n$i.log | tail -1)" | tee -a gpurun_out/r04_suite_repeat.txt
done
bash tools/gpu_soak_r04.sh 1 9000 > /dev/null
tail -40 gpurun_out/r04_soak_9000.log
