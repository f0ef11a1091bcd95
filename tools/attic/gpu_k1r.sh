#!/bin/bash
# front-end tile size experiment: smaller tiles leave room for more front-end workgroups beside the DC / clock stages' LDS
export PYTHONUNBUFFERED=1
set +e
for v in "" "-DSDRM_K1_R=7" "-DSDRM_K1_R=9" "-DSDRM_K1_R=11"; do
    echo "== EXTRA=$v"
    touch sdr-modem_amd/csrc/sdrm_kernels.h
    make -C sdr-modem_amd/csrc EXTRA="$v" > /tmp/k1v_build.log 2>&1 || { tail -5 /tmp/k1v_build.log; continue; }
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|error" | tail -2
    timeout 100 python tools/stage_times.py 256 2>&1 | grep channels
    for c in 256 2048 4096; do timeout 200 python tools/sweep_point.py $c 2>&1 | grep "^channels"; done
done
