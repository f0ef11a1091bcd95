#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference's own test data (build container only).

What this does (and nothing else):
  * copies the DATA files the reference's tests hold for the fsk_demod path
    (test/resources/*.cf32 / *.s8, cited per file below) next to this script;
  * pulls the inline known-answer arrays (numbers only) out of the reference's
    unit tests into ref_unit_vectors.json, recording file:line provenance;
  * emits the two numeric tables the path is defined by (MMSE interpolator taps,
    fast-atan table) into tables.json so tests can check the generated headers.

It never copies source text.  /root/reference does not exist on the GPU box, so
the outputs of this script are committed.
"""
import json
import os
import re
import shutil
import struct
import sys

REF = os.environ.get("SDRM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

# data fixtures: (reference file, what pins it)
DATA = {
    "lucky7.expected.cf32": "test/test_fsk_demod.c:67-79 input (48000,4800,5000,2,2000)",
    "lucky7.expected.s8": "test/test_fsk_demod.c:69 expected, dc on",
    "lucky7.expected.nodc.s8": "test/test_fsk_demod.c:77 expected, dc off",
    "nusat.cf32": "test/test_fsk_demod.c:55 input (192000,40000,5000,1,2000,dc)",
    "processed.s8": "test/test_fsk_demod.c:55 expected",
    "inputnan.cf32": "test/test_fsk_demod.c:63 input (240000,9600,5000,1,2000,dc)",
    "nan.s8": "test/test_fsk_demod.c:63 expected",
}

NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)"


def c_float_arrays(path):
    """Return {name: (line, [floats])} for every `float name[..] = {...};` in a C file."""
    text = open(path).read()
    out = {}
    for m in re.finditer(r"float\s+(\w+)\s*\[\s*\d*\s*\]\s*=\s*\{([^}]*)\}", text):
        name, body = m.group(1), m.group(2)
        line = text.count("\n", 0, m.start()) + 1
        vals = [float(x) for x in re.findall(NUM, re.sub(r"[fF](?=\s*[,}\s]|$)", "", body))]
        out[name] = (line, vals)
    return out


def mmse_table(path):
    text = open(path).read()
    m = re.search(r"float\s+taps\s*\[129\]\[8\]\s*=\s*\{(.*?)\};", text, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    tab = []
    for r in rows:
        vals = [x for x in re.findall(NUM, re.sub(r"(?<=\d)[fF]", "", r))]
        tab.append([float(v) for v in vals[:8]])
    assert len(tab) == 129 and all(len(r) == 8 for r in tab), (len(tab),)
    return tab


def atan_table(path):
    arrs = c_float_arrays(path)
    line, vals = arrs["fast_atan_table"]
    assert len(vals) == 257
    return vals


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present at %s: goldens are committed, nothing to do" % REF)
    prov = {}
    for name, why in DATA.items():
        shutil.copyfile(os.path.join(REF, "test/resources", name), os.path.join(HERE, name))
        os.chmod(os.path.join(HERE, name), 0o644)
        prov[name] = "test/resources/%s -- %s" % (name, why)

    unit = {}

    def grab(cfile, names, extra):
        arrs = c_float_arrays(os.path.join(REF, "test", cfile))
        for n in names:
            line, vals = arrs[n]
            unit["%s:%s" % (cfile, n)] = {"source": "test/%s:%d" % (cfile, line), "values": vals, **extra}

    grab("test_lpf_taps.c", ["expected_taps"],
         {"call": "create_low_pass_filter(1.0, 8000, 1750, 500)", "tolerance": "int(x*1e4) equal"})
    grab("test_lpf.c", ["expected", "expected2"], {})  # regex keeps the LAST duplicate name; handled below
    grab("test_quadrature_demod.c", ["expected", "expected2"],
         {"call": "quadrature_demod_create(25.4, 2000); process(ramp 0..199 complex) in 2 + 198", "tolerance": 1e-3})
    grab("test_dc_blocker.c", ["expected"],
         {"call": "dc_blocker_create(32); process(ramp 0..199)", "tolerance": 1e-3})
    grab("test_clock_recovery_mm.c", ["expected", "expected2"],
         {"call": "clock_mm_create(2.0, 0.25*0.175*0.175, 0.005, 0.175, 0.005, 100); process(ramp) 42 then 36",
          "tolerance": 1e-3})

    # test_lpf.c declares `expected`/`expected2` twice (complex test, then float test): split by position.
    text = open(os.path.join(REF, "test/test_lpf.c")).read()
    found = []
    for m in re.finditer(r"float\s+(\w+)\s*\[\s*\]\s*=\s*\{([^}]*)\}", text):
        line = text.count("\n", 0, m.start()) + 1
        vals = [float(x) for x in re.findall(NUM, re.sub(r"[fF](?=\s*[,}\s]|$)", "", m.group(2)))]
        found.append((m.group(1), line, vals))
    assert [f[0] for f in found] == ["expected", "expected2", "expected", "expected2"], [f[0] for f in found]
    del unit["test_lpf.c:expected"], unit["test_lpf.c:expected2"]
    labels = [("complex_call1", "lpf_create(1,48000,4800,2000,2000,cf32); process(complex ramp[0:250])", 1e-2),
              ("complex_call2", "... process(complex ramp[250:500])", 1e-2),
              ("float_call1", "lpf_create(2,48000,4800,2000,2000,f32); process(ramp[0:500])", 1e-3),
              ("float_call2", "... process(ramp[500:1000])", 1e-3)]
    for (name, line, vals), (lab, call, tol) in zip(found, labels):
        unit["test_lpf.c:" + lab] = {"source": "test/test_lpf.c:%d" % line, "values": vals, "call": call,
                                     "tolerance": tol}
    unit["test_lpf.c:small_buffer"] = {
        "source": "test/test_lpf.c:25-45",
        "call": "lpf_create(2,48000,4800,2000,2000,cf32); feed complex ramp 0,1,1,1 samples",
        "output_lens": [0, 1, 0, 1], "last_value": [-0.005327, -0.007783], "tolerance": 1e-3}
    unit["test_mmse_fir_interpolator.c:normal"] = {
        "source": "test/test_mmse_fir_interpolator.c:10-16", "call": "interp(ramp 0..7, mu=0.14)",
        "values": [3.140217], "tolerance": 1e-3}
    unit["test_clock_recovery_mm.c:small_buffers"] = {
        "source": "test/test_clock_recovery_mm.c:25-42", "call": "feed ramp 0,4,3,1 samples",
        "output_lens": [0, 0, 0, 1]}
    unit["test_sig_source.c:success"] = {
        "source": "test/test_sig_source.c:8-18", "call": "sig_source_create(1.0, 4, 4); process(freq=1, n=4)",
        "values": [1, 0, 0, 1, -1, 0, 0, -1], "tolerance": 1e-2}
    # the file source's frequency offset (src/sdr/file_source.c:120-128): "tx.cf32" is what test_success wrote before
    # (`buffer`, five complex samples), read back through a source created with an offset of 1000 Hz at 48 kHz
    fs_arrays = c_float_arrays(os.path.join(REF, "test/test_file_source.c"))
    unit["test_file_source.c:rx_offset"] = {
        "source": "test/test_file_source.c:47-61 (expected, line %d; input `buffer`, line %d)" % (fs_arrays["expected"][0], fs_arrays["buffer"][0]),
        "call": "file_source_create(1, 'tx.cf32', NULL, 48000, 1000, 2000); sdr_process_rx -> sig_source_multiply(1000, input)",
        "offset_hz": 1000, "sampling_freq": 48000, "input": fs_arrays["buffer"][1], "values": fs_arrays["expected"][1],
        "tolerance": 1e-4}
    unit["perf_fsk_modem.c:config"] = {
        "source": "test/perf_fsk_modem.c:70-98",
        "call": "fsk_demod_create(48000,4800,5000,2,2000,true,2016000); input re=(uint8)i, im=0, 4096 samples; 10x100 calls"}

    with open(os.path.join(HERE, "ref_unit_vectors.json"), "w") as f:
        json.dump({"provenance": prov, "vectors": unit}, f, indent=1)

    tabs = {
        "mmse_source": "src/dsp/mmse_fir_interpolator.c:23-154 (GNU Radio gr-filter interpolator_taps.h, NTAPS 8, NSTEPS 128)",
        "mmse": mmse_table(os.path.join(REF, "src/dsp/mmse_fir_interpolator.c")),
        "atan_source": "src/math/fast_atan2f.c:23-67 (GNU Radio fast_atan2f table, atan(i/255) printed %.6e)",
        "atan": atan_table(os.path.join(REF, "src/math/fast_atan2f.c")),
    }
    with open(os.path.join(HERE, "tables.json"), "w") as f:
        json.dump(tabs, f)
    print("wrote", len(DATA), "data files,", len(unit), "unit vectors, 2 tables")


if __name__ == "__main__":
    main()
