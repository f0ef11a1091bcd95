#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
timeout 600 python -u -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_k.log 2>&1; echo "suite exit $?"; grep -E "passed|failed" gpurun_out/r04_pytest_k.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_soak_r04.sh 0.6 5000
