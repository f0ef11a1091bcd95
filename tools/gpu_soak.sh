#!/bin/bash
# Soak session (since round 5 its fuzz draws low deviations every second round and blocking calls take the in-call hand-off; round 6:
# the hand-off is admitted per device, so the soaks' concurrent handles and batches meet the ledger) -- each soak under its own
# timeout: differential, bit-exact against the oracle or they stop.
# Arguments: seconds scale (1 = the full session of ~35 minutes), first seed offset, round tag (default r06)
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
cd "$R"
mkdir -p gpurun_out
S=${1:-1}
F=${2:-0}
RND=${3:-r06}
O=gpurun_out/${RND}_soak_$F.log
: > $O
git_rev=$(cat .git_rev 2>/dev/null)
echo "$RND soak session (scale $S, seeds +$F), commit ${git_rev:-unknown}" >> $O
run() { echo "\$ $*" >> $O; timeout $1 "${@:2}" 2>&1 | tr "\r" "\n" | grep -a -o "[a-z ]*soak ok:.*\|MISMATCH.*\|HANG.*\|Traceback.*\|Error.*\|failed.*" | tail -3 >> $O; echo "exit ${PIPESTATUS[0]}" >> $O; }
t() { python3 -c "print(int($1 * $S))"; }
run $(t 700) python tools/soak_fuzz.py $(t 600) $((110000 + F))
run $(t 300) python tools/soak_nco.py $(t 200) $((900 + F))
run $(t 300) python tools/soak_batcher.py $(t 200) $((40000 + F))
export SDRM_BATCHER_CALIBRATE=1 SDRM_AUTOTUNE=2  # the calibrated batch behind a batcher (what found the stale hand-off stamps in round 5); every dimension timed
run $(t 200) python tools/soak_batcher.py $(t 120) $((90000 + F))
unset SDRM_BATCHER_CALIBRATE SDRM_AUTOTUNE
run $(t 300) python tools/soak_misc.py $(t 200) $((40000 + F))
run $(t 400) python tools/soak_workers.py $(t 300) $((40000 + F))
run $(t 300) python tools/soak_live.py $(t 200) $((40000 + F))
run $(t 300) python tools/soak_handles.py $(t 200) $((60000 + F))
cat $O
