#!/bin/bash
# Second round-2 soak session on the final commit (each under its own timeout).
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r02_soak_b.log
: > $O
run() { echo "\$ $*" >> $O; timeout $1 "${@:2}" 2>&1 | grep -v "amdgpu.ids\|dsp_worker" | tail -3 >> $O; echo "exit ${PIPESTATUS[0]}" >> $O; }
run 1400 python tools/soak_fuzz.py 1200 80000
run 700 python tools/soak_nco.py 600 500
run 400 python tools/soak_batcher.py 300 20000
run 400 python tools/soak_misc.py 300 20000
cat $O
