#!/bin/bash
# usage (on the gpurun box): tools/gpu_handoff.sh <round, e.g. r05>
# The in-call hand-off's evidence, before / after on one box (SDRM_HANDOFF=0 / 1): a blocking 256 x 131072 call's time and device
# timeline (checked against the oracle), per-role cycles of the DC and clock stages in such a call, one plain handle's
# fsk_demod_process latency over sizes, and the bench line at 20 and 256 steps.  -> gpurun_out/<round>_handoff_raw.txt
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run this on the gpurun box}
RND=${1:-r05}
cd "$R"
mkdir -p gpurun_out
O=gpurun_out/${RND}_handoff_raw.txt
: > $O
echo "commit $(cat .git_rev 2>/dev/null)" >> $O
for h in 0 1; do
  echo "== blocking call, SDRM_HANDOFF=$h" >> $O
  SDRM_HANDOFF=$h SDRM_TIMELINE=1 SDRM_VERIFY=1 timeout 300 python tools/blocking_call.py 256 131072 30 2>&1 | grep -a "^device\|^  [0-9]\|^blocking\|^channel" >> $O
  echo "== stage roles in a blocking call, SDRM_HANDOFF=$h" >> $O
  SDRM_HANDOFF=$h BLOCKING=1 timeout 300 python tools/k3_probe.py 256 2>&1 | grep -a "^wave 0\|^staging\|^K2\|^K1" >> $O
done
echo "== other blocking shapes" >> $O
for shape in "64 131072" "256 32768" "1024 131072" "30 131072"; do
  for h in 0 1; do SDRM_HANDOFF=$h timeout 300 python tools/blocking_call.py $shape 20 2>&1 | grep -a "^blocking" >> $O; done
done
echo "== one plain handle" >> $O
for n in 4096 8192 16384 32768 65536 131072; do
  for h in 0 1; do SDRM_HANDOFF=$h timeout 300 python tools/latency.py $n 100 2>&1 | grep -a "fsk_demod_process" >> $O; done
done
echo "== bench line (ms per step; per-stage kernel ms)" >> $O
for h in 0 1; do for k in 20 256; do
  SDRM_HANDOFF=$h timeout 600 python bench.py --steps $k --no-cpu-baseline --no-extras --sweep "" 2>/dev/null | python3 -c "
import json, sys
j = json.loads(sys.stdin.readline())
print('handoff $h steps $k: %.4f ms per step, %.1f %s' % (j['ms_per_step'], j['value'], j['unit']), j.get('kernel_ms'))" >> $O
done; done
cat $O
