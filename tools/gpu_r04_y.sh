#!/bin/bash
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:?}
mkdir -p gpurun_out
MIX=1 FIRST=200 timeout 300 python tools/timeline_first_calls.py 256 131072 131072 > gpurun_out/r04_timeline_mix.txt 2>&1
tail -64 gpurun_out/r04_timeline_mix.txt
