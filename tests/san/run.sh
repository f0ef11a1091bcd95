#!/bin/bash
# Builds the host-side code (queue.c, batcher.cpp, node.cpp, the planner and the kernel emulation the CPU tests drive) with the
# sanitizers and runs (1) the threaded stress driver, (2) the CPU test-suite's emulation / batcher tests with the
# sanitized emulation library preloaded.  CPU build only.  Logs -> profiles/<round>_sanitizers.txt when called as `run.sh record <round>` (e.g. r06).
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=build/san
mkdir -p $OUT
SRC="tests/san/host_stress.cpp sdr-modem_amd/host/ledger.cpp tests/emu/sdrm_emu.cpp tests/emu/emu_batcher.cpp sdr-modem_amd/host/batcher.cpp sdr-modem_amd/host/node.cpp sdr-modem_amd/csrc/sdrm_plan.cpp sdr-modem_amd/csrc/sdrm_design.cpp"
LIBSRC="tests/emu/sdrm_emu.cpp tests/emu/emu_batcher.cpp sdr-modem_amd/host/batcher.cpp sdr-modem_amd/host/node.cpp sdr-modem_amd/csrc/sdrm_plan.cpp sdr-modem_amd/csrc/sdrm_design.cpp"
FLAGS="-O1 -g -mfma -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-omit-frame-pointer -pthread -Wno-unknown-pragmas"
LOG=$OUT/log.txt
: > $LOG
status=0
for san in "address,undefined" "thread"; do
  tag=$(echo $san | tr ',' '_')
  echo "== -fsanitize=$san" | tee -a $LOG
  gcc -O1 -g -std=gnu11 -fPIC -fsanitize=$san -fno-omit-frame-pointer -pthread -c sdr-modem_amd/host/queue.c -o $OUT/queue_$tag.o || status=1
  extra=""; [ "$san" = "thread" ] && extra="-DSDRM_TSAN_BUILD"
  g++ $FLAGS $extra -fsanitize=$san $SRC $OUT/queue_$tag.o -o $OUT/host_stress_$tag -lm || status=1
  if [ "$san" = "thread" ]; then
    TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" timeout 300 $OUT/host_stress_$tag 2>&1 | grep -v "queue is full" >> $LOG; rc=${PIPESTATUS[0]}
  else
    ASAN_OPTIONS="detect_leaks=1" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" timeout 300 $OUT/host_stress_$tag 2>&1 | grep -v 'queue is full' >> $LOG; rc=${PIPESTATUS[0]}
  fi
  echo "host_stress exit $rc" | tee -a $LOG
  [ $rc -ne 0 ] && status=1
done
# the CPU suite's emulation-backed tests with an ASan+UBSan build of the emulation library (python itself is not instrumented)
g++ $FLAGS -fsanitize=address,undefined -shared $LIBSRC -o $OUT/libsdrm_emu_asan.so -lm || status=1
echo "== pytest (emulation + batcher tests) on the ASan/UBSan build" | tee -a $LOG
SDRM_EMU_LIB=$ROOT/$OUT/libsdrm_emu_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS="detect_leaks=0" \
  UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" timeout 900 python -m pytest tests/test_batcher_cpu.py tests/test_node_cpu.py tests/test_kernel_logic_cpu.py -q -x -p no:cacheprovider >> $LOG 2>&1; rc=$?
echo "pytest exit $rc" | tee -a $LOG
[ $rc -ne 0 ] && status=1
grep -cE "ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer" $LOG | sed 's/^/sanitizer reports: /' | tee -a $LOG
if [ "${1:-}" = "record" ]; then
  { echo "tests/san/run.sh  ($(date -u +%Y-%m-%d), gcc $(gcc -dumpversion), CPU build)"; grep -E "^==|exit|host_stress:|passed|failed|sanitizer reports|ERROR|WARNING: Thread|runtime error" $LOG; } > profiles/${2:-r06}_sanitizers.txt
fi
exit $status
