#!/usr/bin/env python3
"""Run-to-run determinism of the batch engine: the same batch N times, every output compared bit for bit with the
first.  usage: AUDIOMOD_PV_FUSED=2 tools/determinism_probe.py [runs]   (prints mismatching runs with positions)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiomod_amd import engine as E, signals  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
x = np.stack([signals.voice(40000, 2, seed=23 + s) for s in range(3)])
bad_total = 0
for kw in (dict(semitones=4.0), dict(mode="robotic", fftsize=1024), dict(semitones=12.0),
           dict(semitones=-7.0, mode="formant_pitchshift"), dict(mode="time_stretch", time_ratio=0.6, flush=False, coremode=0)):
    kw = dict(kw)
    flush = kw.pop("flush", True)
    xin = torch.from_numpy(x).cuda()
    ref = None
    bad = 0
    for i in range(runs):
        b = E.Batch(3, 40000, channels=2, flush=flush, **kw)   # a fresh engine each time, as the test has
        o = b.run(xin)
        torch.cuda.synchronize()
        o = o.cpu().numpy()
        b.close()
        if ref is None:
            ref = o
        elif not np.array_equal(ref.view(np.uint32), o.view(np.uint32)):
            m = np.argwhere(ref.view(np.uint32) != o.view(np.uint32))
            bad += 1
            print("MISMATCH", kw, "run", i, len(m), m[:4].tolist(), [(float(ref[tuple(j)]), float(o[tuple(j)])) for j in m[:4]])
    print(kw, "runs", runs, "mismatching", bad)
    bad_total += bad
print("total mismatching runs", bad_total)
