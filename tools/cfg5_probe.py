import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from audiomod_amd import engine as E, signals
from oracle import oracle_py as O
S, F = 128, 60 * 48000
kw = dict(mode="normal_pitchshift", semitones=4.0, coremode=1, fftsize=2048)
d_in = signals.synthetic_batch(torch, S, F, torch.device("cuda", 0))
b = E.Batch(S, F, channels=2, block=480, flush=True, **kw)
out = b.run(d_in); torch.cuda.synchronize()
out2 = b.run(d_in); torch.cuda.synchronize()
print("rerun identical:", signals.batch_checksum(torch, out)[0] == signals.batch_checksum(torch, out2)[0])
bad = []
for s in (0, 17, 42, 69, 100, 126):
    want, _, _ = O.run_offline(d_in[s].cpu().numpy(), block=480, flush=True, **kw)
    got = out[s].cpu().numpy()
    d = got.astype(np.float64) - want
    e = float(np.sqrt(np.mean(d ** 2)))
    first = int(np.argmax(np.abs(d).max(axis=0) > 1e-3)) if e > 1e-4 else -1
    print(s, e, "first bad frame", first, flush=True)
