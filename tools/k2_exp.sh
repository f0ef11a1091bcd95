for e in 0 1; do
  touch sdr-modem_amd/csrc/sdrm_kernels.hip
  make -C sdr-modem_amd/csrc EXTRA="-DSDRM_K3_LOOP_SKEW=$e" 2>&1 | grep -E "error" | head -3
  echo "== SKEW=$e"; python tools/k3_ab.py 256 2>&1 | tail -2; python tools/k3_ab.py 4096 2>&1 | tail -2 | head -1
done
