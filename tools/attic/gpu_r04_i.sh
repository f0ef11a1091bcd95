#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
timeout 300 python -u -m pytest tests/test_gpu_fuzz.py -v -x --timeout 200 --timeout-method=thread 2>&1 | grep -E "PASSED|FAILED|passed|failed|Error" | tail -8
timeout 300 python tools/config5_beside.py > gpurun_out/r04_config5_beside.txt 2>&1; cat gpurun_out/r04_config5_beside.txt | grep -v amdgpu.ids
