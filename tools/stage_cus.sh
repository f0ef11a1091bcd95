#!/bin/bash
# usage (on the gpurun box): tools/stage_cus.sh <round>
# Experiment: chain stages (DC blocker, clock recovery) and front-end on disjoint CUs (SDRM_STAGE_CUS="a,b": chains on the
# first a CUs of every XCD, front-end on CUs b..31), at the channel counts where the front-end crowds the chains.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run this on the gpurun box}
RND=${1:-r05}
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/${RND}_stage_cus_raw.txt
: > $O
for ch in 512 1024 2048 4096; do
  for sp in "" "2,2" "4,4" "6,6" "8,8" "12,12" "4,0" "8,0"; do
    echo "== channels $ch SDRM_STAGE_CUS='$sp'" >> $O
    if [ -z "$sp" ]; then timeout 300 python tools/sweep_point.py $ch 2>&1 | head -1 >> $O
    else SDRM_STAGE_CUS=$sp timeout 300 python tools/sweep_point.py $ch 2>&1 | head -1 >> $O; fi
  done
done
cat $O
