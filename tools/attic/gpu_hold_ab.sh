#!/bin/bash
# front-end hold (waits for the clock stage's placement): where should its lower channel bound be?
export PYTHONUNBUFFERED=1
run() { timeout 600 python bench.py --no-cpu-baseline --no-extras --sweep "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"; }
for r in 1 2; do
echo "== bench 256, hold 128..1024 (default)"; run
echo "== bench 256, hold off"; SDRM_FRONT_HOLD=100000,100000 run
done
for c in 128 192 256 384 512; do
  for h in "128,1024" "100000,100000"; do
    echo "== $c channels hold $h: $(SDRM_FRONT_HOLD=$h timeout 300 python tools/sweep_point.py $c 2>&1 | grep '^channels')"
  done
done
