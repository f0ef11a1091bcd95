#!/bin/bash
# Experiment driver (GPU box): rebuild the library with front-end kernel switches and time the stages.
# usage: tools/k1_variants.sh "<EXTRA flags>" ["<EXTRA flags>" ...]
set +e
for v in "$@"; do
    echo "== EXTRA=$v"
    touch sdr-modem_amd/csrc/sdrm_kernels.h
    make -C sdr-modem_amd/csrc EXTRA="$v" > /tmp/k1v_build.log 2>&1 || { tail -5 /tmp/k1v_build.log; continue; }
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|error" | tail -2
    for c in 256 1024; do timeout 100 python tools/stage_times.py $c 2>&1 | grep channels; done
    timeout 100 python tools/k3_probe.py 256 2>&1 | grep "K1 per"
done
