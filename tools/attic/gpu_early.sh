#!/bin/bash
# clock stage of the next call resident early (SDRM_K3_EARLY=<channels>, 0: never): parity, then A/B of the step time
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
for c in 256 512 1024 1280; do
  for e in 0 1280; do
    echo "== $c channels, SDRM_K3_EARLY=$e"
    SDRM_K3_EARLY=$e timeout 300 python tools/sweep_point.py $c 2>&1 | grep "^channels"
  done
done
for e in 0 1280; do
  echo "== config 5, SDRM_K3_EARLY=$e"
  SDRM_K3_EARLY=$e timeout 300 python tools/config5.py 2>&1 | tail -1
  SDRM_K3_EARLY=$e timeout 300 python tools/c5_probe2.py arena c5 2>&1 | tail -1
done
SDRM_K3_EARLY=1280 timeout 300 python tools/sweep_point.py 256 2>&1 | tail -9
