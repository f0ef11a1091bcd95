#!/bin/bash
# round 4, session G: the GPU suite several times over (does any test hang or fail sporadically?)
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
for rep in 1 2 3 4 5 6; do
  timeout 600 python -u -m pytest tests -m gpu -v -x --timeout 150 --timeout-method=thread > gpurun_out/r04_rep_$rep.log 2>&1
  rc=$?
  echo "rep $rep exit $rc: $(grep -E "passed|failed" gpurun_out/r04_rep_$rep.log | tail -1)"
  if [ $rc -ne 0 ]; then
    grep -vE "dsp_worker (is|stopped)" gpurun_out/r04_rep_$rep.log | tail -120 | cut -c1-250
    break
  fi
done
