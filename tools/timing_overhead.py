import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from audiomod_amd import engine as E
S, secs = 128, 20
frames = secs*48000
dev = torch.device('cuda',0)
kw = dict(mode="normal_pitchshift", semitones=4.0, coremode=1, fftsize=2048)
x = bench.make_input(torch, S, frames, dev, 0)
b = E.Batch(S, frames, channels=2, **kw)
o = b.alloc_out()
def t(n=5):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): b.run(x, o)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for _ in range(2): b.run(x,o)
print("no timing ", [round(t(),2) for _ in range(3)])
b.enable_timing(True)
print("timing    ", [round(t(),2) for _ in range(3)])
b.kernel_times(); b.enable_timing(False)
print("no timing ", [round(t(),2) for _ in range(3)])
