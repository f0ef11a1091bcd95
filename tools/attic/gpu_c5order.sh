#!/bin/bash
# does config 5's figure in the bench line depend on what ran before it in the process?
export PYTHONUNBUFFERED=1
for s in "" "512,1024,4096"; do
  echo "== sweep '$s'"
  timeout 600 python bench.py --no-cpu-baseline --sweep "$s" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config5']['value'], d['config5']['kernel_ms'], d['end_to_end']['value'])"
done
timeout 300 python tools/config5.py 2>&1 | grep -v amdgpu.ids | tail -1
