#!/bin/bash
# round 4, session B: the whole GPU suite (node front door, rx offset, advisor fixes included)
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r04_pytest_b.log 2>&1; echo "pytest exit $?"; grep -vE "dsp_worker (is|stopped)" gpurun_out/r04_pytest_b.log | tail -25
