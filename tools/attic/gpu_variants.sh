#!/bin/bash
# Experiment driver (GPU box): rebuild the library with compile-time switches and measure pipelined sweep points.
# usage: tools/gpu_variants.sh <out-name> "<channel counts>" "<EXTRA flags>" ["<EXTRA flags>" ...]
set +e
R=${GRAFT_REPO_ROOT:?}
cd "$R"
OUT="$R/gpurun_out/$1"; shift
CHS=$1; shift
mkdir -p "$(dirname "$OUT")"
export TMPDIR=/tmp
for v in "$@"; do
    echo "== EXTRA=$v" | tee -a "$OUT"
    touch sdr-modem_amd/csrc/sdrm_kernels.h
    make -C sdr-modem_amd/csrc EXTRA="$v" > /tmp/variant_build.log 2>&1 || { tail -5 /tmp/variant_build.log | tee -a "$OUT"; continue; }
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|error" | tail -2 | tee -a "$OUT"
    for c in $CHS; do timeout 200 python tools/sweep_point.py $c 2>/dev/null | head -1 | tee -a "$OUT"; done
done
