"""Throughput of the batch engine at other FFT sizes (not BASELINE configs): 128 stereo streams x 20 s, +4 semitones,
phase-locked; a few streams checked against the oracle.  usage (GPU box): python tools/fft_size_bench.py [fftsize ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiomod_amd import engine as E, signals  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192]
S, F = 128, 20 * 48000
x = signals.synthetic_batch(torch, S, F, torch.device("cuda", 0), 0, duplicates=[])
for n in sizes:
    kw = dict(mode="normal_pitchshift", semitones=4.0, coremode=1, fftsize=n)
    b = E.Batch(S, F, channels=2, **kw)
    out = b.alloc_out()
    for _ in range(2):
        b.run(x, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 4
    for _ in range(steps):
        b.run(x, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    want, _, _ = O.run_offline(x[3].cpu().numpy(), **kw)
    rms = float(np.sqrt(np.mean((out[3].cpu().numpy().astype(np.float64) - want) ** 2)))
    print(f"fft {n:5d}: {S * 2 * F / dt / 1e6:9.1f} Msamples/s  {dt * 1e3:7.2f} ms per pass  rms vs oracle {rms:.2e}", flush=True)
    b.close()
