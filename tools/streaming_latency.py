#!/usr/bin/env python3
"""Latency / throughput of the drop-in streaming API (pv_feed + pv_retrieve, host buffers, PCIe inclusive):
one stereo 48 kHz stream fed in 480-frame calls like audiomod-exe.  Writes a JSON summary."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from audiomod_amd import engine as E, signals  # noqa: E402

secs = 30
x = np.tile(signals.voice(10 * 48000, 2), (1, secs // 10))
res = {}
for block in (480, 4800):
    pv = E.PhaseVocoder(48000, 2, 1.0, 4.0, E.NORMAL_SHIFT, E.PHASE_LOCKED, 2048)
    lat = []
    t0 = time.perf_counter()
    for i in range(0, x.shape[1], block):
        t1 = time.perf_counter()
        pv.processInData(x[:, i:i + block])
        pv.getOutData(pv.getOutSamples())
        lat.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    lat = np.array(lat[20:]) * 1e6
    res[f"block_{block}"] = {"x_realtime": round(secs / dt, 1), "call_us_median": round(float(np.median(lat)), 1),
                             "call_us_p99": round(float(np.percentile(lat, 99)), 1), "calls": len(lat)}
    pv.close()
print(json.dumps({"workload": "1 stereo stream, +4 st, fft 2048, phase-locked, host buffers (ctypes caller)", **res}))
