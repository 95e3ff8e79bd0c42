import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from audiomod_amd import engine as E, signals
if len(sys.argv) > 1 and sys.argv[1] != "-":
    E.LIB_PATH = os.path.join(ROOT, "audiomod_amd", "lib", "diag", sys.argv[1], "libaudiomod_pv.so")
# dirty the device memory first: what a suite of earlier tests leaves behind
junk = torch.full((1 << 28,), 52.5256, device="cuda"); del junk; torch.cuda.empty_cache()
junk = torch.full((1 << 29,), float("nan"), device="cuda"); del junk; torch.cuda.empty_cache()
x = np.stack([signals.voice(40000, 2, seed=23 + s) for s in range(3)])
b = E.Batch(3, 40000, channels=2, flush=True, semitones=4.0)
o = b.run(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
o = o.cpu().numpy(); b.close()
np.save(sys.argv[2], o)
print("saved", o.shape, "nan count", int(np.isnan(o).sum()), "max", float(np.nanmax(np.abs(o))))
