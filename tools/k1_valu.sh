#!/bin/bash
# vector instructions per launch and duration of the front-end alone (stages serialised): tools/k1_valu.sh [channels]
export TMPDIR=/tmp
# tools/stage_times.py waits for every call, and counter collection serialises dispatches: with the in-call hand-off a DC or clock
# workgroup could then be started before the front-end it waits for.  The stages are measured one after the other here.
export SDRM_HANDOFF=0
R=${GRAFT_REPO_ROOT:?run this on the gpurun box (GRAFT_REPO_ROOT is the snapshot root)}
CH=${1:-256}
python $R/tools/stage_times.py $CH
cd /tmp
rm -rf "$R/gpurun_out/pmc_k1v"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_k1v -- python3 $R/tools/stage_times.py $CH > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmc_k1v/*/*counter_collection.csv")[0]
a=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="SQ_INSTS_VALU": a[r["Kernel_Name"][:44]].append(float(r["Counter_Value"]))
for k,v in a.items():
    if "sdrm::k" in k: print("%-46s SQ_INSTS_VALU %.1f M" % (k, sum(v)/len(v)/1e6))
PY
