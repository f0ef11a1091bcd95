#!/bin/bash
# A/B of two builds of the library in one box session (ab/libA.so, ab/libB.so, made beforehand): the clock-stage probe and the
# pipelined step at 256 channels, alternating, so that box-to-box and minute-to-minute drift cancels
export PYTHONUNBUFFERED=1
for round in 1 2 3; do
  for v in A B; do
    cp ab/lib$v.so sdr-modem_amd/csrc/libsdrmodem_hip.so
    echo "== $v round $round"
    timeout 200 python tools/k3_probe.py 256 2>&1 | grep -E "^wave 0"
    timeout 200 python tools/sweep_point.py 256 2>&1 | grep "^channels"
  done
done
cp ab/libB.so sdr-modem_amd/csrc/libsdrmodem_hip.so
