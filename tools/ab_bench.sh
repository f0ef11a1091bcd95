#!/bin/bash
# A/B on ONE box: bench.py alternately with two builds of the library (SDRM_LIB_PATH), n rounds each.
# usage: tools/ab_bench.sh <libA> <libB> [rounds] [steps] [warmup]   -> lines "A <ms_per_step> <k1> <k2> <k3>" / "B ..."
A=$1; B=$2; N=${3:-3}; STEPS=${4:-40}; WARM=${5:-10}
for i in $(seq $N); do
  for side in A B; do
    lib=$A; [ $side = B ] && lib=$B
    SDRM_LIB_PATH=$lib python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras --sweep "" 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        j = json.loads(line); k = j.get('kernel_ms', {})
        print('$side', j['ms_per_step'], k.get('front_lpf1_quad_lpf2'), k.get('dc_blocker'), k.get('clock_recovery'), (j.get('steady_state') or {}).get('ms_per_step'))
"
  done
done
