#!/bin/bash
# NCO stage after a change: parity tests that touch it, its cost in front of the demodulator, config 5, a short NCO soak
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -k "nco or doppler or Doppler or fanout or worker" 2>&1 | tail -4
timeout 300 python tools/nco_times.py 256 131072 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/config5.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python tools/soak_nco.py 120 2>&1 | grep -v amdgpu.ids | tail -3
