#!/bin/bash
# clock stage after a change to its loop: parity (all shapes, fuzz), the probe, the stage times at 256/1024/4096 channels
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do timeout 200 python tools/k3_probe.py 256 2>&1 | grep -E "^wave 0|cycles per iteration"; done
for c in 256 1024 4096; do timeout 300 python tools/stage_times.py $c 2>&1 | grep -v amdgpu.ids | tail -1; done
