#!/bin/bash
# Round-4 session Z: configs[4]'s mix with the clock-stage shape forced to 16x512 -- the case in which a setting that wins every
# measurement of the online refinement (blocks and probation) degrades afterwards -- run long enough for the standing guard
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:?}
mkdir -p gpurun_out
O=gpurun_out/r04_config5_guard.txt
: > $O
for rep in 1 2 3; do
  env SDRM_K3_LANES=16x512 SDRM_AUTOTUNE_LOG=1 timeout 300 python tools/config5.py 256 600 2>&1 | grep -E "channels:|refined|fell behind" | sed 's/sdrmodem_hip: //' | tee -a $O
done
env SDRM_AUTOTUNE_LOG=1 timeout 300 python tools/config5.py 256 600 2>&1 | grep -E "channels:|refined|fell behind" | sed 's/sdrmodem_hip: //' | tee -a $O
timeout 900 python -u -m pytest tests -m gpu -x -q --timeout 250 --timeout-method=thread > gpurun_out/r04_pytest_z.log 2>&1; echo "suite exit $?"; grep -E "passed|failed" gpurun_out/r04_pytest_z.log | tail -1
