"""fft 1024 / 512 through the wave-per-frame kernels (round 3, four-pass WF<512> / WF<256>): every mode, streaming and
batch API, both arithmetic settings, against the oracle.  usage (GPU box): python tools/fft1024_check.py [fftsize=1024]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiomod_amd import engine as E, signals  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

FFT = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = signals.voice(30000, 2, seed=77)
cases = [dict(semitones=4.0), dict(semitones=2.0, coremode=0), dict(semitones=2.0, coremode=2), dict(semitones=-7.0),
         dict(mode="time_stretch", time_ratio=1.5, flush=False), dict(mode="formant_pitchshift", semitones=5.0),
         dict(mode="gender_change", semitones=-4.0), dict(mode="robotic"), dict(mode="constant"), dict(mode="whisper"),
         dict(mode="vocoder"), dict(mode="formant_cepstral", semitones=3.0)]
bad = 0
for arith in (E.ARITH_FAST, E.ARITH_EXACT):
    E.set_arithmetic(arith)
    for kw in cases:
        kw = dict(kw, fftsize=FFT)
        flush = kw.pop("flush", True)
        want, wc, _ = O.run_offline(x, flush=flush, **kw)
        got, gc = E.run_offline(x, flush=flush, **kw)
        r1 = float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2))) if got.shape == want.shape else float("inf")
        S = 5
        b = E.Batch(S, x.shape[1], channels=2, flush=flush, **kw)
        o = b.run(torch.from_numpy(np.stack([x] * S)).cuda())
        torch.cuda.synchronize()
        o = o.cpu().numpy()
        b.close()
        r2 = float(np.sqrt(np.mean((o[S - 1].astype(np.float64) - want) ** 2))) if o[S - 1].shape == want.shape else float("inf")
        same = bool(np.array_equal(o[0].view(np.uint32), got.view(np.uint32)))
        ok = gc == wc and r1 <= 1e-4 and r2 <= 1e-4 and same
        bad += not ok
        print("fast " if arith == E.ARITH_FAST else "exact", kw, f"stream rms {r1:.2e} batch rms {r2:.2e} counts {gc == wc} batch==stream {same}",
              "ok" if ok else "FAIL", flush=True)
print("failed:", bad)
sys.exit(1 if bad else 0)
