#!/bin/bash
# Round-4 session Q: a front-end whose workgroups walk along their channel (SDRM_K1_PERSIST = workgroups per channel), with the
# present 256-thread workgroups and with one-wave workgroups (no barrier couples waves on different SIMDs).
# columns of a cell: ms per step, Msamples/s, front-end / DC / clock stage ms per launch (pipelined)
set +e
R=${GRAFT_REPO_ROOT:?}
OUT="$R/gpurun_out/r04_k1_persist.txt"
export TMPDIR=/tmp
cd "$R"
: > $OUT
cell() { label=$1; shift; r=$(env "$@" timeout 300 python tools/sweep_cell.py $ch 131072 2>/dev/null | tail -1); printf "  %5d ch  %-34s %s\n" $ch "$label" "$r" | tee -a "$OUT"; }
alone() { r=$(env "$@" timeout 200 python tools/stage_times.py $ch 2>&1 | grep channels | tail -1); printf "  serialised: %s\n" "$r" | tee -a "$OUT"; }
variant() {
  echo "== build EXTRA=$1" | tee -a $OUT
  touch sdr-modem_amd/csrc/sdrm_kernels.h
  make -C sdr-modem_amd/csrc EXTRA="$1" > /tmp/k1v_build.log 2>&1 || { tail -5 /tmp/k1v_build.log | tee -a $OUT; return; }
  SDRM_K1_PERSIST=2 timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke|Error|error" | tail -2 | tee -a $OUT
  shift
  for ps in "$@"; do
    for ch in 256 1024 4096; do
      cell "workgroups per channel: $ps" SDRM_K1_PERSIST=$ps
    done
    ch=256; alone SDRM_K1_PERSIST=$ps
  done
}
variant "" 0 16 8 4 0
variant "-DSDRM_K1_THREADS=64 -DSDRM_K1_WGS=16" 0 48 24 12 6
variant "-DSDRM_K1_THREADS=128 -DSDRM_K1_WGS=8" 0 24 12 6
touch sdr-modem_amd/csrc/sdrm_kernels.h
make -C sdr-modem_amd/csrc > /dev/null 2>&1
