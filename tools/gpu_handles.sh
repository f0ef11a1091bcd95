#!/bin/bash
# usage (on the gpurun box): tools/gpu_handles.sh <round, e.g. r06>
# Many private handles / workers calling at once with the reference's buffer size (src/dsp_worker.c:188, :75, config.conf:11):
# tools/handles_bench with the in-call hand-off (admitted per device, sdrm_handoff_stats) and with SDRM_HANDOFF=0, alternately on
# one box.  -> gpurun_out/<round>_handles_raw.txt
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
for shape in "1 131072 20" "8 131072 20" "32 131072 20" "64 131072 20" "128 131072 20" "256 131072 20" "64 32768 40" "256 4096 50"; do
  for rep in 1 2; do
    for h in 1 0; do
      SDRM_HANDOFF=$h timeout 300 tools/handles_bench -q $shape 2>&1 | grep -a "handles x\|NO\|<3>" | head -5 >> $O
    done
  done
done
echo "== workers (dsp_worker_create, private handle each, file sink)" >> $O
for shape in "32 131072 20" "64 131072 20"; do
  for h in 1 0; do
    SDRM_HANDOFF=$h timeout 300 tools/handles_bench -q -w $shape 2>&1 | grep -a "workers x\|NO\|<3>" | head -5 >> $O
  done
done
cat $O
