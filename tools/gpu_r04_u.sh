#!/bin/bash
# Round-4 session U: BASELINE configs[4]'s mix with the schedule's dimensions forced, next to the self-calibrated default
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r04_config5_schedule.txt
: > $O
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/config5.py $ch 2>/dev/null | tail -1 | sed 's/.*channels: //'); printf "  %4d ch  %-36s %s\n" $ch "$label" "$r" | tee -a $O; }
for ch in 256 1024; do
  for rep in 1 2; do
    cell "default (self-calibrated)" SDRM_AUTOTUNE_LOG=1
    cell "calibration off" SDRM_AUTOTUNE=0
    cell "front hold forced on" SDRM_FRONT_HOLD=1,100000
    cell "front hold forced off" SDRM_FRONT_HOLD=0,0
    cell "companion grid forced on" SDRM_K3_COMPANY=4096,1,100000
    cell "companion grid forced off" SDRM_K3_COMPANY=0,0,0
    cell "clock stage 16x512" SDRM_K3_LANES=16x512
    cell "clock stage 32x512" SDRM_K3_LANES=32x512
  done
done
SDRM_AUTOTUNE_LOG=1 timeout 300 python tools/config5.py 256 2>&1 | grep calibrated | tee -a $O
