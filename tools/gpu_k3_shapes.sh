#!/bin/bash
# usage (on the gpurun box): tools/gpu_k3_shapes.sh <round>   -- the 1024-channel sweep point under forced clock-stage shapes.
# A 16 x 1024 clock-stage workgroup holds 141 KB of LDS: no front-end workgroup (37 KB) fits beside it, so at 1024 channels the
# front-end works on 192 of the 256 CUs.  Shapes with shorter rings leave room: does the step gain?  -> gpurun_out/<round>_k3_shapes_1024.txt
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run this on the gpurun box}
RND=${1:-r06}
cd "$R"
O=gpurun_out/${RND}_k3_shapes_1024.txt
: > $O
for rep in 1 2; do
  for shape in default 16 16x512 16x256 32 32x256 64x256p; do
    for ch in 1024 1536; do
      if [ $shape = default ]; then unset SDRM_K3_LANES; else export SDRM_K3_LANES=$shape; fi
      echo -n "shape $shape: " >> $O
      SDRM_AUTOTUNE=0 timeout 200 python tools/sweep_point.py $ch 2>&1 | grep -a "^channels" >> $O
    done
  done
done
unset SDRM_K3_LANES
cat $O
