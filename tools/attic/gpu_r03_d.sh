#!/bin/bash
# round 3, call D: A/B of the discriminator forms on one box: instructions per launch and duration of the front-end alone
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r03d"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
for rep in 1 2; do
  for q in fast flat; do
    echo "== quad $q" | tee -a "$OUT/ab.txt"
    SDRM_K1_QUAD=$q bash tools/k1_valu.sh 256 2>/dev/null | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
  done
done
