#!/bin/bash
# Round-4 session S: the N-rank path of bench.py with 4 and 8 ranks sharing the box's one GPU over gloo (functional, not a
# measurement: the pool has one GPU per box and RCCL needs one per rank)
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp SDRM_BENCH_BACKEND=gloo
for n in 4 8; do
  timeout 900 python bench.py --gpus $n --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_gloo$n.json 2> gpurun_out/r04_bench_gloo$n.err
  echo "ranks $n: exit $?"; cut -c1-200 gpurun_out/r04_bench_gloo$n.json; tail -2 gpurun_out/r04_bench_gloo$n.err
done
