#!/bin/bash
# round 4, session C: (1) MFMA-products + shift-register-adds microbenchmark; (2) the 1024-channel point, round-2 head
# against HEAD against HEAD without the int8 conversion in the staging wave, alternating, same box, same session
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== ubench_mfma_fir"
timeout 300 tools/ubench_mfma_fir > gpurun_out/r04_ubench_mfma_fir.txt 2>&1; echo "exit $?"; cat gpurun_out/r04_ubench_mfma_fir.txt
O=gpurun_out/r04_1024_ab.txt
: > $O
export SDRM_AUTOTUNE=0   # the A/B legs compare fixed schedules; the self-calibration gets its own lines at the end
echo "tools/sweep_cell.py <channels> 131072: ms per step, Msamples/s, kernel ms front / dc / clock" >> $O
for rep in 1 2 3 4; do
  for lib in r02 head nofuse; do
    if [ $lib = head ]; then unset SDRM_LIB_PATH; else export SDRM_LIB_PATH=$R/build/ab/libsdrmodem_$lib.so; fi
    echo "rep $rep $lib 1024: $(timeout 200 python tools/sweep_cell.py 1024 131072 2>&1 | tail -1)" >> $O
  done
done
unset SDRM_LIB_PATH
for hold in "0,0" "128,1024" "832,1024" "512,1280"; do
  echo "head hold=$hold 1024: $(SDRM_FRONT_HOLD=$hold timeout 200 python tools/sweep_cell.py 1024 131072 2>&1 | tail -1)" >> $O
  echo "nofuse hold=$hold 1024: $(SDRM_LIB_PATH=$R/build/ab/libsdrmodem_nofuse.so SDRM_FRONT_HOLD=$hold timeout 200 python tools/sweep_cell.py 1024 131072 2>&1 | tail -1)" >> $O
done
for ch in 256 512 768 1280 2048 4096; do
  for lib in r02 head nofuse; do
    if [ $lib = head ]; then unset SDRM_LIB_PATH; else export SDRM_LIB_PATH=$R/build/ab/libsdrmodem_$lib.so; fi
    echo "$lib $ch: $(timeout 200 python tools/sweep_cell.py $ch 131072 2>&1 | tail -1)" >> $O
  done
done
unset SDRM_LIB_PATH
unset SDRM_AUTOTUNE
echo "--- HEAD with the self-calibration on (what a batch decides for itself)" >> $O
for ch in 256 512 768 1024 1280 2048 4096; do
  echo "head calibrated $ch: $(SDRM_AUTOTUNE_LOG=1 timeout 200 python tools/sweep_cell.py $ch 131072 2>&1 | grep -E "calibrated|^[0-9]" | tr '\n' ' ')" >> $O
done
cat $O
