#!/bin/bash
# Round-4 session L: the tree after the plan-time LDS check: suite, smoke, bench line, worker surface end to end with and
# without the node front door.
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
timeout 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method=thread > gpurun_out/r04_pytest_l.log 2>&1; echo "suite exit $?"; grep -E "passed|failed" gpurun_out/r04_pytest_l.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r04_bench_l.json 2> gpurun_out/r04_bench_l.err; echo "bench exit $?"
gcc -O2 -pthread tools/batcher_bench.c -Iinclude -Lsdr-modem_amd/csrc -lsdrmodem_hip -Wl,-rpath,'$ORIGIN/../sdr-modem_amd/csrc' -lm -o tools/batcher_bench || exit 1
{
  for mode in 0 1 2 4; do
    for rep in 1 2; do
      timeout 300 tools/batcher_bench 256 131072 16 8 4 $mode 2>&1 | grep -v "^<"
    done
  done
  timeout 300 tools/batcher_bench 1024 131072 8 8 4 0 2>&1 | grep -v "^<"
  timeout 300 tools/batcher_bench 1024 131072 8 8 4 4 2>&1 | grep -v "^<"
} > gpurun_out/r04_batcher_node.txt
cat gpurun_out/r04_batcher_node.txt
