#!/usr/bin/env python3
"""Known-answer vectors of the reference's cepstral formant shift (formantShiftSlice, phasevocoderprocess.cc:925-999,
dead code upstream: called directly by oracle/_ref/ref_formant, which links the compiled reference) at the FFT sizes
tools/make_golden.py does not cover: 256, 512, 1024, 8192.  Data only: input magnitudes, envelope factor, output.
Needs oracle/_ref (build container, where /root/reference exists).  -> tests/golden/kat_formant_sizes.npz"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402


def main():
    rng = np.random.default_rng(20260204)
    exe = os.path.join(os.path.dirname(O.REF_DRIVER), "ref_formant")
    kat = {}
    with tempfile.TemporaryDirectory() as d:
        fm, fo = os.path.join(d, "fm.f32"), os.path.join(d, "fo.f32")
        for N in (256, 512, 1024, 8192):
            H = N // 2 + 1
            k = np.arange(H)
            env = np.exp(-((k - 0.06 * N) / (0.04 * N)) ** 2) + 0.6 * np.exp(-((k - 0.2 * N) / (0.06 * N)) ** 2) + 0.05
            voiced = (env * (1 + 0.8 * np.cos(2 * np.pi * k / 9.3)) * 50 + rng.random(H)).astype(np.float32)
            mags = np.stack([voiced, voiced[::-1].copy(), (rng.random(H) * 100).astype(np.float32),
                             np.zeros(H, np.float32), (rng.random(H) ** 8 * 1e-3).astype(np.float32)]).astype(np.float32)
            mags.tofile(fm)
            kat[f"formant{N}_in"] = mags
            for tag, e in (("+5", 2.0 ** (np.float32(5.0) / np.float32(12.0))),
                           ("-9", 2.0 ** (np.float32(-9.0) / np.float32(12.0))), ("1", 1.0)):
                e = float(np.float32(e))
                subprocess.run([exe, str(N), repr(e), fm, fo], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                kat[f"formant{N}_{tag}_env"] = np.array([e], np.float32)
                kat[f"formant{N}_{tag}_out"] = np.fromfile(fo, np.float32).reshape(-1, H)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "kat_formant_sizes.npz"), **kat)
    print({k: v.shape for k, v in kat.items()})


if __name__ == "__main__":
    main()
