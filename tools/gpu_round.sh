#!/bin/bash
# One GPU-box session: smoke, parity tests, bench, rocprofv3 kernel stats.  Logs -> gpurun_out/.
# Every step runs under its own `timeout` so that a wedged device cannot hold the box.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
echo "== bench"
timeout 600 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.log
echo "== rocprof"
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --sweep "" > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -6 $f; done
echo "== host-buffer paths"
timeout 200 python tools/host_pipeline.py 256 131072 24
timeout 200 tools/batcher_bench 256 131072 16 8 4
