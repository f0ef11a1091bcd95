#!/bin/bash
# The schedule under a server's real load (review of round 4, item 5): a node over ONE 512-slot batcher serving 3 / 30 / 256 live
# clients of BASELINE configs[4]'s three kinds, 131072-sample buffers (src/resources/config.conf:11); the batcher's own schedule
# (calibrated at creation on 512 placeholder channels, refined online) against every forced setting.  -> ms per round.
# usage (GPU box): tools/node_schedule.sh [rounds]
R=${1:-24}
run() { # label, env...
  label=$1; shift
  for n in 3 30 256; do
    out=$(env "$@" BB_SLOTS=512 BB_KINDS=3 timeout 300 tools/batcher_bench $n 131072 $R 8 4 1 2>/dev/null | grep "end to end" | sed -E 's/.*: ([0-9.]+) ms per round, ([0-9]+) Msamples.*/\1 \2/')
    printf "%-46s %4d clients: %8s ms per round, %6s Msamples/s\n" "$label" $n $out
  done
}
run "batcher's own (calibration + online refinement)"
run "rules only (SDRM_AUTOTUNE=0)" SDRM_AUTOTUNE=0
for shape in 16x1024 32x512 64x256p; do
  for comp in "0,0,0" "4096,0,100000,1"; do
    for hold in "100000,100001" "0,100000"; do
      run "forced $shape company ${comp%%,*} hold $([ $hold = 0,100000 ] && echo on || echo off)" SDRM_AUTOTUNE=0 SDRM_K3_LANES=$shape SDRM_K3_COMPANY=$comp SDRM_FRONT_HOLD=$hold
    done
  done
done
