#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
timeout 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method=thread -k "any_samples_per_symbol" 2>&1 | tail -3
bash tools/gpu_r04_grid.sh > /dev/null
wc -l gpurun_out/r04_heuristics_grid.txt
