"""Round-3 diagnosis of the rare fused / tile mismatch: what does the FIRST configuration of
tests/test_gpu_parity.py::test_fused_overlap_add_is_bit_identical_to_the_tile_path put out when run 0 of a row starts
from the final accumulator image of a complete pass instead of from zeros?  Needs a -DPV_DIAG build
(tools/build_variant.sh diag -DPV_DIAG) selected with AUDIOMOD_PV_LIB, AUDIOMOD_PV_FUSED=2 and
AUDIOMOD_PV_DEBUG_STALE_ACC=1.  Round 2's failing runs showed 52.56 where 2.578e-4 belongs at [2, 1, 2]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiomod_amd import engine as E, signals  # noqa: E402

x = np.stack([signals.voice(40000, 2, seed=23 + s) for s in range(3)])
for kw in (dict(semitones=4.0), dict(mode="robotic", fftsize=1024)):
    b = E.Batch(3, 40000, channels=2, flush=True, **kw)
    xin = torch.from_numpy(x).cuda()
    first = b.run(xin).cpu().numpy()   # the image is zeros at creation: this pass is the regular one
    second = b.run(xin).cpu().numpy()  # this one starts from what the first pass left behind
    m = np.argwhere(first.view(np.uint32) != second.view(np.uint32))
    print(kw, "differing samples:", len(m))
    for s in range(3):
        for c in range(2):
            print("  row", s, c, "first", first[s, c, :4], "second", second[s, c, :4])
    if len(m):
        print("  first positions:", m[:6].tolist(), "last:", m[-3:].tolist())
    b.close()
