for e in 0; do
  touch sdr-modem_amd/csrc/sdrm_kernels.hip
  make -C sdr-modem_amd/csrc EXTRA="-DSDRM_K2_DEBUG_STAGE -DSDRM_K2_EXP=$e" 2>&1 | grep -E "error" | head -3
  echo "== EXP=$e"; timeout 300 python tools/k3_probe.py 256 2>&1 | tail -2 | head -1
done
