#!/bin/bash
export PYTHONUNBUFFERED=1
run() { timeout 600 python bench.py --no-cpu-baseline --no-extras --sweep "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"; }
echo "== early on, company off"; SDRM_K3_EARLY=1000 SDRM_K3_COMPANY=0,0,0 run
echo "== early off, company off"; SDRM_K3_EARLY=0 SDRM_K3_COMPANY=0,0,0 run
echo "== early on, front hold off"; SDRM_FRONT_HOLD=100000,100000 run
echo "== early off, front hold off"; SDRM_K3_EARLY=0 SDRM_FRONT_HOLD=100000,100000 run
echo "== early on"; run
SDRM_K3_EARLY=1000 timeout 300 python tools/sweep_point.py 256 2>&1 | tail -9
