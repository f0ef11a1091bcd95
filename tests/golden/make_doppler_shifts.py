#!/usr/bin/env python3
"""Regenerate tests/golden/doppler_shifts_lucky7.json (build container only): the per-second Doppler shifts of the
reference's own Doppler test case (test/test_doppler.c:14,38: LUCKY-7 TLE, station 53.72N 47.57E, 437.525 MHz,
start 1583840449, 48 kHz), computed by oracle/_ref/ref_doppler_shifts = the reference's vendored SGP4 sources."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(HERE, "..", "..", "oracle", "_ref", "ref_doppler_shifts")
TLE = ["LUCKY-7", "1 44406U 19038W   20069.88080907  .00000505  00000-0  32890-4 0  9992",
       "2 44406  97.5270  32.5584 0026284 107.4758 252.9348 15.12089395 37524"]
args = ["53.72", "47.57", "0.0", "48000", "437525000", "0", "1583840449", "8"]
# the reference passes float literals 53.72F / 47.57F / 0.0F promoted to double (test_doppler.c:38)
import struct
f32 = lambda v: struct.unpack("<f", struct.pack("<f", v))[0]
args[0], args[1] = repr(f32(53.72)), repr(f32(47.57))
out = subprocess.check_output([TOOL] + args + TLE).decode()
shifts = json.loads(out)
json.dump({"source": "test/test_doppler.c:14-42 case; src/dsp/doppler.c:31-42 formula on the reference's SGP4 (src/sgpsdp)",
           "sampling_freq": 48000, "center_freq": 437525000, "start_time": 1583840449, "shifts_hz": shifts},
          open(os.path.join(HERE, "doppler_shifts_lucky7.json"), "w"), indent=1)
print(shifts)
