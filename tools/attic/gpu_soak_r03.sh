#!/bin/bash
# Round-3 soak session (each soak under its own timeout): differential, bit-exact against the oracle or they stop.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
cd "$R"
mkdir -p gpurun_out
O=gpurun_out/r03_soak.log
: > $O
git_rev=$(cat .git_rev 2>/dev/null)
echo "round-3 soak session, commit ${git_rev:-unknown}" >> $O
run() { echo "\$ $*" >> $O; timeout $1 "${@:2}" 2>&1 | tr "\r" "\n" | grep -a -o "[a-z ]*soak ok:.*\|MISMATCH.*\|Traceback.*\|Error.*" | tail -3 >> $O; echo "exit ${PIPESTATUS[0]}" >> $O; }
run 1300 python tools/soak_fuzz.py 1200 90000
run 500 python tools/soak_nco.py 400 700
run 400 python tools/soak_batcher.py 300 30000
run 400 python tools/soak_misc.py 300 30000
run 300 python tools/soak_workers.py 200 30000
run 300 python tools/soak_live.py 200 30000
cat $O
