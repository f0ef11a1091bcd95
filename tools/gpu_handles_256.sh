#!/bin/bash
# usage (on the gpurun box): tools/gpu_handles_256.sh -- 256 private handles x 131072-sample buffers, hand-off admitted by the device's
# ledger vs SDRM_HANDOFF=0, five alternating rounds (ten-second runs of 256 threads scatter by +-5 %), then 1 and 2 handles
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
gcc -O2 -pthread tools/handles_bench.c -Iinclude -Lsdr-modem_amd/csrc -lsdrmodem_hip -Wl,-rpath,$GRAFT_REPO_ROOT/sdr-modem_amd/csrc -lm -o tools/handles_bench || exit 1
O=gpurun_out/r06_handles_256_raw.txt
: > $O
for rep in 1 2 3 4 5; do
  for h in 1 0; do
    SDRM_HANDOFF=$h timeout 300 tools/handles_bench -q -W 4 256 131072 24 2>&1 | grep -a "handles x\|NO\|<3>" | head -3 >> $O
  done
done
for n in 1 2; do
  for rep in 1 2; do
    for h in 1 0; do
      SDRM_HANDOFF=$h timeout 300 tools/handles_bench -q -W 4 $n 131072 24 2>&1 | grep -a "handles x\|NO\|<3>" | head -3 >> $O
    done
  done
done
cat $O
