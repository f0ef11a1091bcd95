#!/bin/bash
# round 4, session H: generic channels on the device, then the whole suite
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
timeout 600 python -u -m pytest tests/test_gpu_parity.py -v -x -k "any_samples" --timeout 200 --timeout-method=thread > gpurun_out/r04_generic.log 2>&1; echo "generic exit $?"
grep -vE "dsp_worker (is|stopped)" gpurun_out/r04_generic.log | tail -40 | cut -c1-250
timeout 900 python -u -m pytest tests -m gpu -v -x --timeout 200 --timeout-method=thread > gpurun_out/r04_pytest_h.log 2>&1; echo "suite exit $?"
grep -E "passed|failed" gpurun_out/r04_pytest_h.log | tail -2
grep -E "FAILED|Error|Timeout" gpurun_out/r04_pytest_h.log | head -10
