#!/bin/bash
# round 4, session A: the bench-shape parity tests and the bench line with its spot checks
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench-shape parity"
timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -x --timeout 900 > gpurun_out/r04_shapes.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r04_shapes.log
echo "== bench (driver's flags)"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_a.json 2>gpurun_out/r04_bench_a.err; echo "bench exit $?"; cat gpurun_out/r04_bench_a.json; tail -3 gpurun_out/r04_bench_a.err
