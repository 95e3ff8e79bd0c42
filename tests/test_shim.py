"""The drop-in C++ class audiomod::phasevocoder (include/dafx/phasevocoder.h) driven by a stand-alone
program with the reference CLI's two loops, compared with the golden vectors of the real reference."""
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import ROOT, load_e2e

pytestmark = pytest.mark.gpu
DEMO = os.path.join(ROOT, "audiomod_amd", "lib", "shim_demo")
MODES = {"normal_pitchshift": 0, "gender_change": 1, "formant_pitchshift": 2, "time_stretch": 5, "robotic": 6}


@pytest.mark.parametrize("name", ["cfg2_shift+4_cm1_stereo", "cfg3_stretch1.5_cm1_4096", "cfg4_gender-7",
                                  "rt_shift+4_cm1"])
def test_cpp_class_matches_reference(name, tmp_path):
    assert os.path.exists(DEMO), "run `make` first"
    x, y, counts, meta = load_e2e(name)
    fin, fout, fcnt = (str(tmp_path / n) for n in ("in.f32", "out.f32", "cnt.txt"))
    x.tofile(fin)
    cmd = [DEMO, meta.get("api", "offline"), fin, fout, fcnt, str(x.shape[0]), str(x.shape[1]), "48000",
           repr(float(meta.get("time_ratio", 1.0))), repr(float(meta.get("semitones", 0.0))),
           str(MODES[meta.get("mode", "normal_pitchshift")]), str(meta.get("coremode", 1)),
           str(meta.get("fftsize", 2048)), "480", "1" if meta.get("flush", True) else "0"]
    subprocess.run(cmd, check=True, timeout=120)
    vals = [int(v) for v in open(fcnt).read().split()]
    assert vals[0] == y.shape[1]
    assert vals[1:] == counts
    got = np.fromfile(fout, np.float32).reshape(x.shape[0], vals[0])
    assert float(np.sqrt(np.mean((got.astype(np.float64) - y) ** 2))) <= 1e-4
