#!/bin/bash
# Round-4 session AA: the companion grid started only once the clock stage's workgroups are placed
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:?}
mkdir -p gpurun_out
O=gpurun_out/r04_company_after_placement.txt
: > $O
echo "== configs[4]'s mix, 256 channels, clock stage forced, companion grid forced on / off (tools/config5.py)" | tee -a $O
for sh in 16x512 16x1024 32x512; do for co in "4096,1,100000" "0,0,0"; do for early in 0 1; do
  if [ $early = 1 ]; then e="SDRM_COMPANY_EARLY=1"; else e="A=1"; fi
  r=$(env $e SDRM_K3_LANES=$sh SDRM_K3_COMPANY=$co timeout 300 python tools/config5.py 256 60 2>&1 | grep "channels:" | sed 's/.*channels: //')
  printf "  %-8s grid %-14s %-22s %s\n" $sh $co "$([ $early = 1 ] && echo 'at once (before)' || echo 'after placement')" "$r" | tee -a $O
done; done; done
echo "== the headline workload (tools/sweep_cell.py <channels> 131072): ms per step, Msamples/s, kernel ms" | tee -a $O
for ch in 64 256 512; do for rep in 1 2; do for early in 0 1; do
  if [ $early = 1 ]; then e="SDRM_COMPANY_EARLY=1"; else e="A=1"; fi
  r=$(env $e timeout 300 python tools/sweep_cell.py $ch 131072 2>/dev/null | tail -1)
  printf "  %4d ch  %-22s %s\n" $ch "$([ $early = 1 ] && echo 'at once (before)' || echo 'after placement')" "$r" | tee -a $O
done; done; done
