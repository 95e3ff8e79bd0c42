import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from audiomod_amd import engine as E, signals
from oracle import oracle_py as O
S, F = 128, 60 * 48000
kw = dict(mode="normal_pitchshift", semitones=4.0, coremode=1, fftsize=2048)
d_in = signals.synthetic_batch(torch, 70, F, torch.device("cuda", 0))
x = d_in[69].cpu().numpy()
for start in (203 * 4330, 203 * 4380, 203 * 4400, 203 * 4410):
    seg = np.ascontiguousarray(x[:, start:start + 40000])
    want, _, _ = O.run_offline(seg, block=480, flush=True, **kw)
    got, _ = E.run_offline(seg, block=480, flush=True, **kw)
    d = got.astype(np.float64) - want
    e = float(np.sqrt(np.mean(d ** 2)))
    first = int(np.argmax(np.abs(d).max(axis=0) > 1e-3)) if e > 1e-4 else -1
    print("start", start, "rms", e, "first bad", first, flush=True)
    if e > 1e-4:
        np.save("gpurun_out/r02/repro_seg.npy", (seg * 32768).astype(np.int16))
        break
