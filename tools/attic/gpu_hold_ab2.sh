#!/bin/bash
export PYTHONUNBUFFERED=1
for n in 131072 32768; do
for c in 512 640 768 896 1024; do
  for h in "128,1024" "100000,100000"; do
    echo "chunk $n x $c channels, hold $h: $(SDRM_FRONT_HOLD=$h timeout 300 python tools/sweep_cell.py $c $n 2>/dev/null | tail -1)"
  done
done
done
