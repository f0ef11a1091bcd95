#!/bin/bash
# One GPU-box session: smoke, parity tests, VALU microbench, bench, rocprofv3 kernel stats.  Logs -> gpurun_out/.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== host"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)))"
nproc
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
echo "== ubench"
timeout 300 tools/ubench_valu > gpurun_out/ubench_valu.log 2>&1; cat gpurun_out/ubench_valu.log
echo "== bench"
timeout 900 python bench.py --steps 16 --warmup 4 --verify > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
echo "== rocprof"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sweep "" > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -12 $f; done
