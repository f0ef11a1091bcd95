#!/bin/bash
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:?}
mkdir -p gpurun_out
timeout 300 python tools/timeline_first_calls.py 1024 4096 131072 > gpurun_out/r04_timeline_first.txt 2>&1
cat gpurun_out/r04_timeline_first.txt | tail -70
