#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for args in "256 1 1" "256 1 0" "256 0 1" "256 0 0"; do
  timeout 300 python tools/k3_probe_c5.py $args 2>&1 | grep -v "^<\|amdgpu.ids"
done
} > gpurun_out/r04_k3_probe_c5.txt
cat gpurun_out/r04_k3_probe_c5.txt
