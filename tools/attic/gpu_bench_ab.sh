#!/bin/bash
# the bench line's headline and roofline block with the next call's clock stage resident early (default) and not
export PYTHONUNBUFFERED=1
for r in 1 2; do
for e in "" 0; do
  echo "== SDRM_K3_EARLY='$e'"
  if [ -z "$e" ]; then unset SDRM_K3_EARLY; else export SDRM_K3_EARLY=$e; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --sweep "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['roofline'].get('frac_of_ceiling'))"
done
done
unset SDRM_K3_EARLY
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -3
for c in 512 768; do for e in "" 0; do if [ -z "$e" ]; then unset SDRM_K3_EARLY; else export SDRM_K3_EARLY=$e; fi; echo "== $c early=$e"; timeout 300 python tools/sweep_point.py $c 2>&1 | grep "^channels"; done; done
