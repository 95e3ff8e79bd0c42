import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from audiomod_amd import engine as E, signals
from oracle import oracle_py as O
S, F = 128, 60 * 48000
kw = dict(mode="normal_pitchshift", semitones=4.0, coremode=1, fftsize=2048)
d_in = signals.synthetic_batch(torch, S, F, torch.device("cuda", 0))
x = d_in[69].cpu().numpy()
np.save("gpurun_out/r02/stream69.npy", (x[:, 880000:920000] * 32768).astype(np.int16))
want, _, _ = O.run_offline(x, block=480, flush=True, **kw)
def report(tag, got):
    d = got.astype(np.float64) - want
    e = float(np.sqrt(np.mean(d ** 2)))
    first = int(np.argmax(np.abs(d).max(axis=0) > 1e-3)) if e > 1e-4 else -1
    print(tag, e, "first bad frame", first, flush=True)
b = E.Batch(1, F, channels=2, block=480, flush=True, **kw)
out = b.run(d_in[69:70].contiguous()); torch.cuda.synchronize()
report("batch-of-1", out[0].cpu().numpy())
got, _ = E.run_offline(x, block=480, flush=True, **kw)
report("streaming", got)
for cm in (0, 2):
    kw2 = dict(kw, coremode=cm)
    w2, _, _ = O.run_offline(x, block=480, flush=True, **kw2)
    b2 = E.Batch(1, F, channels=2, block=480, flush=True, **kw2)
    o2 = b2.run(d_in[69:70].contiguous()); torch.cuda.synchronize()
    d = o2[0].cpu().numpy().astype(np.float64) - w2
    print("coremode", cm, float(np.sqrt(np.mean(d ** 2))))
