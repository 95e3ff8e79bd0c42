#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random configurations (mode, pitch, ratio, core mode, FFT size, hop, rate,
channels, length, call size, signal) through the streaming and the batch API against the oracle.  Prints one line
per case and a summary; exit code 1 if any case fails.  usage: tests/sweeps/fuzz_parity.py [cases] [seed] [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audiomod_amd import engine as E  # noqa: E402
from audiomod_amd import signals  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

RMS_TOL = 1e-4


def rms(a, b):
    """RMS difference over the samples that are finite in the oracle's output; the reference emits NaN in places
    (e.g. its Rosenberg carrier below ~22 kHz sample rate: a zero-length opening phase), and there the engine must
    emit non-finite values at the same positions.  Returns inf on a mismatch of those positions."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(b)
    if not np.array_equal(np.isfinite(a), fin):
        return float("inf")
    return float(np.sqrt(np.mean((a[fin] - b[fin]) ** 2))) if fin.any() else 0.0


def draw(rng):
    mode = rng.choice(["normal_pitchshift"] * 4 + ["time_stretch"] * 2 + ["gender_change", "formant_pitchshift",
                      "robotic", "constant", "vocoder", "vocoder_chord", "formant_cepstral", "whisper"])
    kw = dict(mode=str(mode))
    kw["fftsize"] = int(rng.choice([256, 512, 1024, 2048, 2048, 2048, 4096, 4096, 8192]))
    if mode == "formant_cepstral":  # every size since round 2 (the lifter needs 128 points or more)
        kw["fftsize"] = int(rng.choice([256, 512, 1024, 2048, 4096, 8192]))
    kw["coremode"] = int(rng.choice([0, 1, 1, 2]))
    kw["sample_rate"] = int(rng.choice([8000, 16000, 22050, 44100, 48000, 48000, 96000]))
    if mode == "time_stretch":
        kw["time_ratio"] = float(np.float32(rng.choice([rng.uniform(0.3, 3.5), rng.choice([0.5, 1.0, 1.5, 2.0, 3.0])])))
    else:
        st = rng.choice([rng.uniform(-16, 16), rng.uniform(-16, 16), float(rng.integers(-14, 15)),
                         float(rng.integers(-14, 15)), 0.0])
        kw["semitones"] = float(np.float32(st))
    if rng.random() < 0.2:
        kw["hopsize"] = int(rng.integers(16, kw["fftsize"] // 2))
    ch = int(rng.choice([1, 2, 2, 2, 3, 4]))
    frames = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 60000)]))
    block = int(rng.choice([480, 480, 64, 4800, int(rng.integers(2, 6000)), int(rng.integers(2, 6000)), 1]))
    if block < 16 and frames > 4000:
        frames = 4000
    kind = str(rng.choice(["voice", "voice", "noise", "sweep", "burst", "zeros"]))
    flush = bool(mode != "time_stretch") if rng.random() < 0.8 else bool(rng.random() < 0.5)
    return kw, ch, frames, block, kind, flush


def signal(kind, frames, ch, seed):
    if kind == "voice":
        return signals.voice(frames, ch, seed=seed)
    if kind == "noise":
        return signals.noise(frames, ch, seed=seed)
    if kind == "sweep" and frames >= 1000:
        return signals.sweep(frames, ch)
    if kind == "burst" and frames >= 1000:
        return signals.silence_burst(frames, ch, seed=seed)
    if kind == "zeros":
        return np.zeros((ch, frames), np.float32)
    return signals.voice(frames, ch, seed=seed)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad, unsupported, done = [], [], 0
    for i in range(cases):
        if time.time() - t0 > budget:
            break
        kw, ch, frames, block, kind, flush = draw(rng)
        x = signal(kind, frames, ch, 1000 + i)
        tag = f"#{i} {kw} ch={ch} frames={frames} block={block} {kind} flush={flush}"
        try:
            want, wc, _ = O.run_offline(x, block=block, flush=flush, **kw)
        except O.OracleUndefined:
            # undefined behaviour in the reference: the engine has to refuse, not to compute something
            try:
                E.run_offline(x, block=block, flush=flush, **kw)
                print(tag, "reference undefined, engine computed something: FAIL", flush=True)
                bad.append(tag)
            except E.PvError as ex:
                print(tag, "reference undefined, engine refuses:", ex, flush=True)
            continue
        try:
            got, gc = E.run_offline(x, block=block, flush=flush, **kw)
        except E.PvError as ex:
            unsupported.append(tag + f" -> {ex}")
            print(tag, "UNSUPPORTED", ex, flush=True)
            continue
        done += 1
        ok = list(gc) == list(wc) and got.shape == want.shape
        if not ok:
            print("   counts/shape differ:", got.shape, want.shape, list(gc)[:12], list(wc)[:12], flush=True)
        r = rms(got, want) if ok else float("nan")
        ok = ok and r <= RMS_TOL
        # the processBlock / outputReady loop (main/main.cc:561-572) on every third case
        if ok and i % 3 == 1:
            try:
                wr, wrc = O.run_realtime(x, block=block, **kw)
                gr, grc = E.run_realtime(x, block=block, **kw)
                okr = list(grc) == list(wrc) and gr.shape == wr.shape and rms(gr, wr) <= RMS_TOL
                if kw["mode"] == "time_stretch":  # pass-through in the reference: bit for bit
                    okr = okr and np.array_equal(gr.view(np.uint32), x.view(np.uint32))
                if not okr:
                    print("   realtime API differs:", gr.shape, wr.shape, list(grc)[:12], list(wrc)[:12],
                          rms(gr, wr) if gr.shape == wr.shape else "", flush=True)
                ok = ok and okr
            except O.OracleUndefined:
                pass
            except E.PvError as ex:
                unsupported.append(tag + f" [realtime] -> {ex}")
        # batch API, two streams holding different inputs
        rb = float("nan")
        okb = True
        if frames >= 1 and i % 2 == 0:
            import torch
            # 2 ... 24 streams (its own generator: the case sequence stays what it was), every stream its own input
            ns = int(np.random.default_rng(seed * 100003 + i).choice([2, 2, 3, 5, 9, 24])) if frames <= 20000 else 2
            xs = [x] + [signal(kind, frames, ch, 5000 + 37 * i + k) for k in range(1, ns)]
            try:
                b = E.Batch(ns, frames, channels=ch, block=block, flush=flush, **kw)
                out = b.run(torch.from_numpy(np.stack(xs)).cuda())
                torch.cuda.synchronize()
                out = out.cpu().numpy()
                b.close()
                okb = out.shape[2] == want.shape[1]
                rb = float("nan")
                if okb:
                    rb = rms(out[0], want)
                    for k in range(1, ns):
                        wk, _, _ = O.run_offline(xs[k], block=block, flush=flush, **kw)
                        rb = max(rb, rms(out[k], wk))
                okb = okb and rb <= RMS_TOL
            except E.PvError as ex:
                unsupported.append(tag + f" [batch] -> {ex}")
                print(tag, "BATCH UNSUPPORTED", ex, flush=True)
        print(tag, f"rms={r:.2e} batch={rb:.2e}", "ok" if ok and okb else "FAIL", flush=True)
        if not (ok and okb):
            bad.append(tag)
    print(f"{done} cases compared, {len(bad)} failed, {len(unsupported)} unsupported, {time.time() - t0:.0f} s")
    for t in bad:
        print("FAILED:", t)
    for t in unsupported:
        print("UNSUPPORTED:", t)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
