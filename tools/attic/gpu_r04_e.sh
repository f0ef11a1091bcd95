#!/bin/bash
# round 4, session E: round-2 head against HEAD at 512 / 1024 / 4096 channels, alternating, same box, same session;
# then the companion grid's footprint (tools/company_sweep.py)
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp SDRM_AUTOTUNE=0
O=gpurun_out/r04_1024_ab2.txt
: > $O
echo "tools/ab_cell.py <lib> <channels> 131072: ms per step, Msamples/s, kernel ms front / dc / clock (self-calibration off)" >> $O
for rep in 1 2 3 4; do
  for lib in build/ab/libsdrmodem_r02.so sdr-modem_amd/csrc/libsdrmodem_hip.so; do
    echo "rep $rep $lib 1024: $(timeout 200 python tools/ab_cell.py $lib 1024 131072 2>&1 | tail -1)" >> $O
  done
done
for ch in 512 4096 256; do
  for rep in 1 2; do
    for lib in build/ab/libsdrmodem_r02.so sdr-modem_amd/csrc/libsdrmodem_hip.so; do
      echo "rep $rep $lib $ch: $(timeout 200 python tools/ab_cell.py $lib $ch 131072 2>&1 | tail -1)" >> $O
    done
  done
done
cat $O
echo "== company sweep 256"
timeout 600 python tools/company_sweep.py 256 131072 > gpurun_out/r04_company_256.txt 2>&1; cat gpurun_out/r04_company_256.txt
echo "== company sweep 512"
timeout 600 python tools/company_sweep.py 512 131072 > gpurun_out/r04_company_512.txt 2>&1; cat gpurun_out/r04_company_512.txt
