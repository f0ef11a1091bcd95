#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== ubench_k3_wave"
timeout 120 tools/ubench_k3_wave > gpurun_out/r04_ubench_k3_wave.txt 2>&1; echo "exit $?"; cat gpurun_out/r04_ubench_k3_wave.txt
echo "== timing cost"
timeout 200 python tools/timing_cost.py 256 2>&1 | grep timing
echo "== config5 alone (tools/config5.py) vs inside bench's function, same box"
timeout 200 python tools/config5.py 256 2>&1 | grep mixed
timeout 200 python tools/config5.py 256 2>&1 | grep mixed
