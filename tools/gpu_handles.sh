#!/bin/bash
# usage (on the gpurun box): tools/gpu_handles.sh <round, e.g. r06>
# Many private handles / workers calling at once with the reference's buffer size (src/dsp_worker.c:188, :75, config.conf:11):
# tools/handles_bench (4 warm-up + 20 timed buffers per client) with the in-call hand-off (admitted per device,
# sdrm_handoff_stats) and with SDRM_HANDOFF=0, alternately on one box; the same clients behind one shared batcher
# (SDRM_SHARED_SLOTS); more hardware queues (GPU_MAX_HW_QUEUES).  -> gpurun_out/<round>_handles_raw.txt
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run this on the gpurun box}
RND=${1:-r06}
cd "$R"
mkdir -p gpurun_out
O=gpurun_out/${RND}_handles_raw.txt
: > $O
echo "commit $(cat .git_rev 2>/dev/null)" >> $O
gcc -O2 -pthread tools/handles_bench.c -Iinclude -Lsdr-modem_amd/csrc -lsdrmodem_hip -Wl,-rpath,$R/sdr-modem_amd/csrc -lm -o tools/handles_bench || exit 1
line() { grep -a "handles x\|workers x\|NO\|<3>" | head -4 >> $O; }
echo "== private handles, hand-off admitted by the device's ledger vs SDRM_HANDOFF=0 (two runs each)" >> $O
for shape in "1 131072 24" "2 131072 24" "3 131072 24" "4 131072 24" "8 131072 24" "16 131072 24" "64 131072 24" "256 131072 24" "64 32768 44" "256 4096 54"; do
  for rep in 1 2; do
    for h in 1 0; do
      SDRM_HANDOFF=$h timeout 300 tools/handles_bench -q -W 4 $shape 2>&1 | line
    done
  done
done
echo "== private workers (dsp_worker_create, private handle each, file sink)" >> $O
for shape in "32 131072 24" "64 131072 24"; do
  for h in 1 0; do
    SDRM_HANDOFF=$h timeout 300 tools/handles_bench -q -w -W 4 $shape 2>&1 | line
  done
done
echo "== the same clients behind ONE shared batcher (SDRM_SHARED_SLOTS = number of handles)" >> $O
for n in 8 64 256; do
  SDRM_SHARED_SLOTS=$n timeout 300 tools/handles_bench -q -W 4 $n 131072 24 2>&1 | line
done
echo "== more hardware queues for the private handles (GPU_MAX_HW_QUEUES; HIP's default is 4 per priority level), SDRM_HANDOFF=0" >> $O
for q in 2 4 8 16; do
  echo -n "GPU_MAX_HW_QUEUES=$q: " >> $O
  GPU_MAX_HW_QUEUES=$q SDRM_HANDOFF=0 timeout 300 tools/handles_bench -q -W 4 64 131072 14 2>&1 | line
done
cat $O
