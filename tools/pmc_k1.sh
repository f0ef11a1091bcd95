#!/bin/bash
# PMC passes for the pipeline kernels (separate passes; no trace domains besides kernel-trace)
set +e
export TMPDIR=/tmp
# tools/stage_times.py waits for every call, and counter collection serialises dispatches: with the in-call hand-off a DC or clock
# workgroup could then be started before the front-end it waits for.  The stages are measured one after the other here.
export SDRM_HANDOFF=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
run() { # name counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/stage_times.py ${CH:-256} > $OUT/$name.log 2>&1
  echo "$name exit $?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for name in ["sq1","sq2","fetch","write"]:
    files = glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % name, recursive=True)
    if not files: print(name, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0][-24:]
        if "sdrm" not in r["Kernel_Name"]: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, {c: round(sum(v[-6:]) / len(v[-6:]), 1) for c, v in d.items()})
PY
