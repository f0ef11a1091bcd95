#!/bin/bash
# quick GPU session: smoke + selected tests + short bench.  usage: tools/gpu_quick.sh "<pytest -k expr>" [bench args]
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -k "$1" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu.log
shift
timeout 600 python bench.py "$@" > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
