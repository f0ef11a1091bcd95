#!/bin/bash
# Round-2 soaks on the final code (each under its own timeout; bit-exact against the oracle or they stop).
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r02_soak.log
: > $O
run() { echo "\$ $*" >> $O; timeout $1 "${@:2}" 2>&1 | grep -v "amdgpu.ids\|dsp_worker" | tail -3 >> $O; echo "exit ${PIPESTATUS[0]}" >> $O; }
run 800 python tools/soak_fuzz.py 660 40000
run 400 python tools/soak_fuzz.py 300 60000
run 330 python tools/soak_batcher.py 240 9000
run 260 python tools/soak_misc.py 180 9000
run 200 python tools/soak_workers.py 120 9000
run 200 python tools/soak_live.py 120 9000
cat $O
