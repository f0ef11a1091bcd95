#!/bin/bash
# Round-4 session Z: configs[4]'s mix, clock-stage shapes forced, the online refinement deciding hold and grid for each
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:?}
mkdir -p gpurun_out
O=gpurun_out/r04_config5_shapes.txt
: > $O
for rep in 1 2; do
for sh in default 16x1024 16x512 16x256 32x512 32x256; do
  if [ $sh = default ]; then e="A=1"; else e="SDRM_K3_LANES=$sh"; fi
  r=$(env $e SDRM_AUTOTUNE_LOG=1 timeout 300 python tools/config5.py 256 2>&1 | grep -E "channels:|refined" | sed 's/.*channels: //; s/sdrmodem_hip: refined online for calls with NCO batches, [0-9]* samples per call: //' | tr '\n' '|')
  printf "  %-8s %s\n" $sh "$r" | tee -a $O
done
done
