#!/bin/bash
# usage (on the gpurun box): tools/gpu_profiles.sh <round, e.g. r05>
# A round's evidence for profiles/: rocprofv3 kernel stats of the bench command, PMC passes (separate: SQ, FETCH_SIZE,
# WRITE_SIZE) over tools/stage_times.py at 256 and 4096 channels, kernel stats of the 1024 / 4096 sweep points and of the
# mixed-rate Doppler workload.  Everything lands under gpurun_out/<round>/; tools/collect_profiles.py <round> (run at home) copies the summaries into profiles/.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run this on the gpurun box (GRAFT_REPO_ROOT is the snapshot root)}
RND=${1:-r05}
OUT=$R/gpurun_out/$RND
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
# the profiled passes run the schedule's starting point (no calibration launches among the averaged kernels); the plain bench at the
# end runs as a user would (calibration on)
export SDRM_AUTOTUNE=0
echo "== bench under rocprofv3 --kernel-trace --stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 $R/bench.py --no-cpu-baseline --no-extras --sweep "" > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; echo "exit $?"
for ch in 1024 4096; do
  echo "== sweep point $ch under rocprofv3"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sweep_${ch}_stats -- python3 $R/tools/sweep_point.py $ch > $OUT/sweep_$ch.txt 2>&1; echo "exit $?"
done
echo "== config5 under rocprofv3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config5_stats -- python3 $R/tools/config5.py 256 > $OUT/config5.txt 2>&1; echo "exit $?"
pmc() { # tag channels name counters...   (stage_times.py waits for every call: without the in-call hand-off the stages run one after the other)
  tag=$1; ch=$2; name=$3; shift 3
  SDRM_HANDOFF=0 timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${tag}_$name -- python3 $R/tools/stage_times.py $ch > $OUT/pmc_${tag}_$name.log 2>&1
  echo "pmc $tag $name exit $?"
}
for ch in 256 4096; do
  pmc c$ch $ch sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pmc c$ch $ch sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pmc c$ch $ch grbm GRBM_GUI_ACTIVE GRBM_COUNT
  pmc c$ch $ch fetch FETCH_SIZE
  pmc c$ch $ch write WRITE_SIZE
done
cd $R
unset SDRM_AUTOTUNE
echo "== plain bench (full line)"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-400 $OUT/bench.json
