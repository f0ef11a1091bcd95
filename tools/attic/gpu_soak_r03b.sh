#!/bin/bash
# Round-3 soak session on the final kernels (NCO phase without a compare, next call's clock stage resident early -- forced for
# half of the runs with SDRM_K3_EARLY, since the library itself switches it on only for long calls of 32..768 channels)
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
cd "$R"
mkdir -p gpurun_out
O=gpurun_out/r03_soak_b.log
: > $O
git_rev=$(cat .git_rev 2>/dev/null)
echo "round-3 soak session B, commit ${git_rev:-unknown}" >> $O
run() { echo "\$ $*" >> $O; timeout $1 "${@:2}" 2>&1 | tr "\r" "\n" | grep -a -o "[a-z ]*soak ok:.*\|MISMATCH.*\|Traceback.*\|Error.*" | tail -3 >> $O; echo "exit ${PIPESTATUS[0]}" >> $O; }
run 600 python tools/soak_fuzz.py 500 91000
echo "-- SDRM_K3_EARLY=100000 (forced)" >> $O
export SDRM_K3_EARLY=100000
run 600 python tools/soak_fuzz.py 500 92000
run 300 python tools/soak_batcher.py 200 31000
run 300 python tools/soak_misc.py 200 31000
run 250 python tools/soak_workers.py 150 31000
run 250 python tools/soak_live.py 150 31000
unset SDRM_K3_EARLY
echo "-- default" >> $O
run 300 python tools/soak_nco.py 200 701
cat $O
