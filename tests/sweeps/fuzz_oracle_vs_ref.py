#!/usr/bin/env python3
"""Randomised pinning of the oracle against the COMPILED REFERENCE (oracle/_ref/ref_driver; only where
/root/reference exists, i.e. in the build container): random configurations through both, outputs compared bit for
bit (NaN where the reference has NaN), per-call counts equal.  CPU only.
usage: tests/sweeps/fuzz_oracle_vs_ref.py [cases] [seed]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audiomod_amd import signals  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def draw(rng):
    mode = str(rng.choice(["normal_pitchshift"] * 4 + ["time_stretch"] * 2 + ["gender_change", "formant_pitchshift",
                          "robotic", "constant", "vocoder", "vocoder_chord"]))
    kw = dict(mode=mode, fftsize=int(rng.choice([256, 512, 1024, 2048, 2048, 4096, 8192])),
              coremode=int(rng.choice([0, 1, 1, 2])),
              sample_rate=int(rng.choice([8000, 16000, 22050, 44100, 48000, 48000, 96000])))
    if mode == "time_stretch":
        kw["time_ratio"] = float(np.float32(rng.choice([rng.uniform(0.3, 3.5), rng.choice([0.5, 1.0, 1.5, 2.0, 3.0])])))
    else:
        kw["semitones"] = float(np.float32(rng.choice([rng.uniform(-16, 16), float(rng.integers(-14, 15)), 0.0])))
    if rng.random() < 0.25:
        kw["hopsize"] = int(rng.integers(16, kw["fftsize"] // 2))
    ch = int(rng.choice([1, 2, 2, 3, 4]))
    frames = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 40000)]))
    block = int(rng.choice([480, 480, 64, 4800, int(rng.integers(16, 6000))]))
    kind = str(rng.choice(["voice", "noise", "burst", "zeros"]))
    flush = bool(mode != "time_stretch") if rng.random() < 0.8 else bool(rng.random() < 0.5)
    api = "rt" if rng.random() < 0.25 else "offline"
    return kw, ch, frames, block, kind, flush, api


def signal(kind, frames, ch, seed):
    if kind == "noise":
        return signals.noise(frames, ch, seed=seed)
    if kind == "burst" and frames >= 1000:
        return signals.silence_burst(frames, ch, seed=seed)
    if kind == "zeros":
        return np.zeros((ch, frames), np.float32)
    return signals.voice(frames, ch, seed=seed)


def same_bits(a, b):
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32)))


def main():
    if not O.have_ref():
        print("oracle/_ref not built (no /root/reference here): nothing to compare against")
        return 0
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    bad, skipped, done = [], 0, 0
    for i in range(cases):
        kw, ch, frames, block, kind, flush, api = draw(rng)
        x = signal(kind, frames, ch, 100 + i)
        tag = f"#{i} {api} {kw} ch={ch} frames={frames} block={block} {kind} flush={flush}"
        try:
            if api == "rt":
                want, wc = O.run_realtime(x, block=block, **kw)
            else:
                want, wc, _ = O.run_offline(x, block=block, flush=flush, **kw)
        except O.OracleUndefined:
            skipped += 1  # the reference overruns its own buffers there: nothing defined to compare
            continue
        try:
            got, gc = O.ref_run(x, api=api, block=block, flush=flush, timeout=60, **kw)
        except (subprocess.CalledProcessError, subprocess.TimeoutExpired) as ex:
            print(tag, "reference run failed:", type(ex).__name__, flush=True)
            bad.append(tag)
            continue
        done += 1
        ok = list(gc) == list(wc) and same_bits(got, want)
        if not ok:
            print(tag, "DIFFERS", got.shape, want.shape, list(gc)[:8], list(wc)[:8], flush=True)
            bad.append(tag)
    print(f"{done} configurations bit-identical to the compiled reference: {done - len(bad)}; differing or failed: "
          f"{len(bad)}; reference-undefined (skipped): {skipped}; {time.time() - t0:.0f} s")
    for t in bad:
        print("BAD:", t)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
