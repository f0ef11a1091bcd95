export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
gcc -O2 -pthread tools/handles_bench.c -Iinclude -Lsdr-modem_amd/csrc -lsdrmodem_hip -Wl,-rpath,$GRAFT_REPO_ROOT/sdr-modem_amd/csrc -lm -o tools/handles_bench || exit 1
O=gpurun_out/r06_hwq2_raw.txt
: > $O
for n in 4 16 64 256; do
  for spread in 0 1; do
    echo "spread=$spread" >> $O
    if [ $spread = 1 ]; then export SDRM_HANDLE_PRIO_SPREAD=1; else unset SDRM_HANDLE_PRIO_SPREAD; fi
    SDRM_HANDOFF=0 timeout 200 tools/handles_bench -q $n 131072 10 2>&1 | grep -a "handles x\|NO\|<3>" | head -3 >> $O
  done
done
unset SDRM_HANDLE_PRIO_SPREAD
echo "hand-off, few handles" >> $O
for n in 1 2 3 4 6 8; do
  for h in 1 0; do
    SDRM_HANDOFF=$h timeout 200 tools/handles_bench -q $n 131072 20 2>&1 | grep -a "handles x\|NO\|<3>" | head -3 >> $O
  done
done
cat $O
