#!/bin/bash
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:?}
mkdir -p gpurun_out
FIRST=20 timeout 300 python tools/timeline_first_calls.py 256 131072 131072 > gpurun_out/r04_timeline_256.txt 2>&1
sed -n 2,24p gpurun_out/r04_timeline_256.txt
FIRST=20 SDRM_K3_COMPANY=0,0,0 timeout 300 python tools/timeline_first_calls.py 256 131072 131072 2>&1 | sed -n 2,14p
