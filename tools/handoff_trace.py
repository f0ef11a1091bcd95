"""rocprofv3's view of a blocking call with the in-call hand-off: from a --kernel-trace CSV of tools/blocking_call.py, the start and end
of the three stages of the last calls (ns from the front-end's start).  python tools/handoff_trace.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "sdrm::k" in r["Kernel_Name"]]
fronts = [k for k in ks if "k1_front" in k[0]]
print("call: front [start, end]  dc [start, end]  clock [start, end]  (ms from the front-end's start; rocprofv3 --kernel-trace)")
for name, s0, e0 in fronts[-6:]:
    # the DC and clock kernels of the same call: the ones whose interval overlaps or follows this front-end first
    dc = min((k for k in ks if "k2_dc<" in k[0] and k[2] > s0), key=lambda k: k[1], default=None)
    ck = min((k for k in ks if "k3_clock<" in k[0] and k[2] > s0), key=lambda k: k[1], default=None)
    if dc is None or ck is None:
        continue
    f = lambda t: (t - s0) / 1e6
    print("  %-28s [%7.3f, %7.3f]  [%7.3f, %7.3f]  [%7.3f, %7.3f]   %s | %s" %
          (name.split("(")[0].replace("void sdrm::", ""), 0.0, f(e0), f(dc[1]), f(dc[2]), f(ck[1]), f(ck[2]),
           dc[0].split("(")[0].replace("void sdrm::", ""), ck[0].split("(")[0].replace("void sdrm::", "")))
