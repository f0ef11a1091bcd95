#!/bin/bash
# round 4: the whole GPU suite, unbuffered and verbose (a hung test must leave its name behind), every test under a
# thread-method timeout (the watchdog thread ends the process even when the main thread sits in a C call)
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
timeout 1200 python -u -m pytest tests -m gpu -v -x --timeout 150 --timeout-method=thread > gpurun_out/r04_pytest_b.log 2>&1; echo "pytest exit $?"
grep -vE "dsp_worker (is|stopped)" gpurun_out/r04_pytest_b.log | grep -E "PASSED|FAILED|ERROR|Timeout|passed|failed" | tail -100 | cut -c1-200
grep -vE "dsp_worker (is|stopped)" gpurun_out/r04_pytest_b.log | tail -60 | cut -c1-250
