#!/bin/bash
# round 4, session D: find the hang of the node front door's GPU test (unbuffered, verbose, thread-method timeouts)
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
timeout 500 python -u -m pytest tests/test_gpu_node.py -v -x -s --timeout 100 --timeout-method=thread > gpurun_out/r04_node_d.log 2>&1; echo "pytest exit $?"
grep -vE "dsp_worker (is|stopped)" gpurun_out/r04_node_d.log | tail -80 | cut -c1-300
