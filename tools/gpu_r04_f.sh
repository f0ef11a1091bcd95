#!/bin/bash
# round 4, session F: what bounds k1_front once the vector pipe is out of the way -- PMC passes (one per counter group, kernel
# trace only) over tools/stage_times.py at 256 and 4096 channels, exact build and the FMA build (half the filter instructions)
set +e
export TMPDIR=/tmp SDRM_AUTOTUNE=0
R=${GRAFT_REPO_ROOT:?}
OUT=$R/gpurun_out/r04_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
pmc() { # tag channels name counters...
  tag=$1; ch=$2; name=$3; shift 3
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${tag}_$name -- python3 $R/tools/stage_times.py $ch > $OUT/${tag}_$name.log 2>&1
  echo "pmc $tag $name exit $?"
}
for mode in exact fast; do
  if [ $mode = fast ]; then export SDRM_STAGE_FAST=1; else unset SDRM_STAGE_FAST; fi
  for ch in 256 4096; do
    pmc ${mode}_c$ch $ch sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
    pmc ${mode}_c$ch $ch sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
    pmc ${mode}_c$ch $ch sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES
    pmc ${mode}_c$ch $ch grbm GRBM_GUI_ACTIVE GRBM_COUNT
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
out = open("gpurun_out/r04_pmc_k1_summary.txt", "w")
for mode in ("exact", "fast"):
    for ch in (256, 4096):
        row = {}
        dur = None
        for name in ("sq1", "sq2", "sq3", "grbm"):
            files = glob.glob("gpurun_out/r04_pmc/%s_c%d_%s/**/*counter_collection.csv" % (mode, ch, name), recursive=True)
            if not files:
                continue
            agg = collections.defaultdict(list)
            durs = []
            for r in csv.DictReader(open(files[0])):
                if "k1_front" not in r["Kernel_Name"]:
                    continue
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r and "End_Timestamp" in r:
                    durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            for c, v in agg.items():
                row[c] = sum(v[-4:]) / len(v[-4:])
            if durs and name == "grbm":
                dur = sum(durs[-4:]) / len(durs[-4:])
        line = "%s %d: duration_ns %s %s" % (mode, ch, dur, {k: round(v, 1) for k, v in sorted(row.items())})
        print(line)
        out.write(line + "\n")
out.close()
PY
