"""python tools/collect_profiles.py <round>: copies the summaries of tools/gpu_profiles.sh <round> (gpurun_out/<round>/) into
profiles/ (tracked) under that round's names and derives profiles/<round>_k1_front_traffic.json, <round>_kernels.json (what
bench.py's roofline.kernels quotes) and <round>_pmc_summary.txt from the PMC passes."""
import collections, csv, glob, json, os, shutil, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", RND)
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)  # gpurun merges runs: the newest one counts
    return f[-1] if f else None


def copy(pattern, name):
    f = one(pattern)
    if f:
        shutil.copyfile(f, os.path.join(DST, name))
        print("copied", name)
    else:
        print("MISSING", pattern)


def short(kernel):
    return kernel.split("(")[0].replace("void ", "").replace("sdrm::", "")


def counters(tag, name):
    f = one("pmc_%s_%s/*/*counter_collection.csv" % (tag, name))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v[2:]) / max(len(v[2:]), 1) for c, v in d.items()} for k, d in agg.items()}  # first two launches: warm-up


def durations(tag, name):
    f = one("pmc_%s_%s/*/*kernel_trace.csv" % (tag, name))
    d = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v[2:]) / max(len(v[2:]), 1) for k, v in d.items()}


copy("bench.json", RND + "_bench.json")
copy("bench_under_rocprof.json", RND + "_bench_under_rocprof.json")
copy("bench_stats/*/*kernel_stats.csv", RND + "_kernel_stats.csv")
copy("sweep_1024_stats/*/*kernel_stats.csv", RND + "_sweep1024_kernel_stats.csv")
copy("sweep_4096_stats/*/*kernel_stats.csv", RND + "_sweep4096_kernel_stats.csv")
copy("config5_stats/*/*kernel_stats.csv", RND + "_config5_kernel_stats.csv")
for tag in ("c256", "c4096"):
    for name in ("sq1", "sq2", "grbm", "fetch", "write"):
        copy("pmc_%s_%s/*/*counter_collection.csv" % (tag, name), RND + "_pmc_%s_%s.csv" % (tag, name))

lines = ["PMC summary of round " + RND + " (tools/gpu_profiles.sh: tools/stage_times.py <channels>, stages serialised, 131072-sample calls;",
         "means over the launches after the two warm-up calls; one rocprofv3 --pmc pass per counter group).",
         "clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; valu_issue = SQ_INSTS_VALU * 4 cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8);",
         "HBM bytes: FETCH_SIZE (KiB) doubled (gfx950 tallies 128-byte reads as 64, MI355X_MICROARCH.md), WRITE_SIZE (KiB) as is.", ""]
for tag, ch in (("c256", 256), ("c4096", 4096)):
    sq1, sq2, grbm = counters(tag, "sq1"), counters(tag, "sq2"), counters(tag, "grbm")
    fe, wr = counters(tag, "fetch"), counters(tag, "write")
    dur = durations(tag, "grbm")
    lines.append("== %d channels" % ch)
    for k in sorted(sq1):
        if not k.startswith(("k1_front", "k2_dc", "k3_clock")):
            continue
        gui = grbm.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        d = dur.get(k, 0.0)
        valu = sq1[k].get("SQ_INSTS_VALU", 0.0)
        lines.append("%-16s %8.3f ms  clock %.2f GHz  SQ_INSTS_VALU %8.1f M  valu_issue %5.1f %%  SQ_WAVES %7.0f  SQ_INSTS_LDS %7.1f M  "
                     "LDS_BANK_CONFLICT/IDX_ACTIVE %.3f  HBM read %7.1f MB  write %7.1f MB" % (
                         k, d / 1e6, gui / d if d else 0.0, valu / 1e6, 100.0 * valu * 4 / 1024 / gui if gui else 0.0,
                         sq1[k].get("SQ_WAVES", 0.0), sq2.get(k, {}).get("SQ_INSTS_LDS", 0.0) / 1e6,
                         sq2.get(k, {}).get("SQ_LDS_BANK_CONFLICT", 0.0) / max(sq2.get(k, {}).get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0),
                         fe.get(k, {}).get("FETCH_SIZE", 0.0) * 2 * 1024 / 1e6, wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6))
    lines.append("")
    if ch == 256:
        # per kernel, what bench.py's roofline.kernels quotes: time alone on the chip, algorithmic and counted HBM bytes, issue share
        algo = {"k1_front": 256 * 131072 * (8 + 4), "k2_dc": 256 * 131072 * (4 + 4), "k3_clock": 256 * 131072 * 4 + 256 * 26215 * (4 + 1)}
        kj = {"round": RND, "channels": 256, "chunk": 131072, "kernels": {},
              "algorithmic_bytes": "front-end: 8 B of IQ read + 4 B of LPF2 output written per sample; DC blocker: 4 B read + 4 B "
                                   "written; clock stage: 4 B read per sample + 4 B (float) + 1 B (int8) written per symbol",
              "method": "tools/gpu_profiles.sh: rocprofv3 --kernel-trace --pmc <group>, one pass per group, over tools/stage_times.py 256 "
                        "(stages one after the other); FETCH_SIZE doubled for gfx950, WRITE_SIZE as is (KiB)"}
        for k in sorted(sq1):
            stem = "k1_front" if k.startswith("k1_front") else "k2_dc" if k.startswith("k2_dc") and "generic" not in k else \
                   "k3_clock" if k.startswith("k3_clock") and "generic" not in k else None
            if stem is None or not dur.get(k):
                continue
            gui = grbm.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            kj["kernels"][stem] = {"avg_ms_alone": round(dur[k] / 1e6, 4), "algorithmic_bytes": algo[stem],
                                   "counter_bytes": int(fe.get(k, {}).get("FETCH_SIZE", 0.0) * 2 * 1024 + wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024),
                                   "valu_issue": round(sq1[k].get("SQ_INSTS_VALU", 0.0) * 4 / 1024 / gui, 4) if gui else None,
                                   "shader_clock_ghz": round(gui / dur[k], 3)}
        total_algo = 256 * 131072 * 8 + 256 * 26215
        total_counted = sum(v["counter_bytes"] for v in kj["kernels"].values())
        kj["whole_step_traffic_ratio"] = round(total_counted / total_algo, 2) if total_counted else None
        json.dump(kj, open(os.path.join(DST, RND + "_kernels.json"), "w"), indent=1)
        print("wrote kernels.json", kj["whole_step_traffic_ratio"])
        k = [x for x in fe if x.startswith("k1_front")][0]
        rd, wrb = fe[k]["FETCH_SIZE"] * 2 * 1024, wr[k]["WRITE_SIZE"] * 1024
        gui = grbm[k]["GRBM_GUI_ACTIVE"] / 8.0
        tj = {"kernel": "k1_front", "round": RND, "channels": 256, "chunk": 131072,
              "fetch_size_kib_raw": round(fe[k]["FETCH_SIZE"], 1), "write_size_kib_raw": round(wr[k]["WRITE_SIZE"], 1),
              "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wrb), "hbm_bytes_per_launch": int(rd + wrb),
              "algorithmic_read_bytes": 256 * 131072 * 8, "algorithmic_write_bytes": 256 * 131072 * 4,
              "sq_insts_valu": int(sq1[k]["SQ_INSTS_VALU"]), "duration_ms_alone": round(dur[k] / 1e6, 4),
              "shader_clock_ghz": round(gui / dur[k], 3), "valu_issue_busy": round(sq1[k]["SQ_INSTS_VALU"] * 4 / 1024 / gui, 4),
              "source": ["profiles/%s_pmc_c256_%s.csv" % (RND, g) for g in ("fetch", "write", "sq1", "grbm")],
              "method": "tools/gpu_profiles.sh + tools/collect_profiles.py: rocprofv3 --kernel-trace --pmc <group> in separate "
                        "passes over tools/stage_times.py 256, mean over the launches after two warm-up calls; FETCH_SIZE / WRITE_SIZE are "
                        "KiB, FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md, HBM section); shader clock = GRBM_GUI_ACTIVE / 8 XCDs / "
                        "kernel duration; valu_issue_busy = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / those cycles"}
        json.dump(tj, open(os.path.join(DST, RND + "_k1_front_traffic.json"), "w"), indent=1)
        print("wrote k1_front_traffic.json", tj["hbm_bytes_per_launch"], tj["shader_clock_ghz"], tj["valu_issue_busy"])
open(os.path.join(DST, RND + "_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
