#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:?}
export TMPDIR=/tmp
cd "$R"
timeout 900 python -m pytest tests -m gpu -q -x -k "batch_256 or fixtures or ragged or many_channel or soak" 2>&1 | tail -2
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$R/gpurun_out/k3t_$c"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/k3t_$c" -- python3 "$R/tools/stage_times.py" 256 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("$R/gpurun_out/k3t_%s/*/*counter_collection.csv"%c)[0]
    a=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==c: a[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k,v in a.items():
        if "sdrm::k" in k: print("%-42s %s %.1f MB" % (k, c, sum(v[2:])/len(v[2:])*1024*(2 if c=="FETCH_SIZE" else 1)/1e6))
PY
