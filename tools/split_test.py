import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from audiomod_amd import engine as E
S, secs = 128, 20
frames = secs*48000
dev = torch.device('cuda',0)
kw = dict(mode="normal_pitchshift", semitones=4.0, coremode=1, fftsize=2048)
x = bench.make_input(torch, S, frames, dev, 0)
for split in (1, 2, 4):
    bs = [E.Batch(S//split, frames, channels=2, **kw) for _ in range(split)]
    outs = [b.alloc_out() for b in bs]
    streams = [torch.cuda.Stream() for _ in range(split)]
    xs = [x[i*(S//split):(i+1)*(S//split)].contiguous() for i in range(split)]
    def run():
        for b,o,s,xi in zip(bs,outs,streams,xs):
            with torch.cuda.stream(s):
                b.run(xi, o, stream=s)
    for _ in range(2): run()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/5
    print(f"split {split}: {dt*1e3:.2f} ms/step  {S*frames*2/dt/1e6:.0f} Msamples/s")
    del bs, outs
