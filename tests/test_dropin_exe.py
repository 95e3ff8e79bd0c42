"""INTEGRATION.md option A as a build and a run: the reference's own CLI -- its main/main.cc and main/wavfile.cc and
every non-phase-vocoder effect, compiled unchanged from /root/reference by oracle/ref.mk (target `dropin`) -- linked
against THIS repository's audiomod::phasevocoder (include/dafx/phasevocoder.h + audiomod_amd/csrc/phasevocoder.cc over
libaudiomod_pv.so) in place of src/phasevocoder/*.  The binary lives in oracle/_ref (test infrastructure, never
shipped, built only where the reference tree is mounted and carried to the GPU box).

Reference: main/main.cc:170,201-286 (construction), :471-510 (offline loop), CMakeLists.txt:64-72 (what links)."""
import ast
import glob
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import GOLD, ROOT

EXE = os.path.join(ROOT, "oracle", "_ref", "audiomod-exe-mi355x")
CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLD, "wav_*.npz")))


def _have():
    return os.path.exists(EXE)


@pytest.mark.skipif(not _have(), reason="oracle/_ref/audiomod-exe-mi355x not built (needs /root/reference)")
def test_dropin_exe_links_the_engine_not_the_reference_core():
    syms = subprocess.run(["nm", "-C", EXE], capture_output=True, text=True, check=True).stdout
    assert " U pv_create" in syms and " U pv_feed" in syms and " U pv_retrieve" in syms
    assert "phasevocodercore" not in syms  # nothing of src/phasevocoder/* is in the binary
    assert "audiomod::phasevocoder::processInData" in syms  # the class main.cc calls: ours
    needed = subprocess.run(["readelf", "-d", EXE], capture_output=True, text=True, check=True).stdout
    assert "libaudiomod_pv.so" in needed


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_reference_cli_over_the_dropin_class(name, tmp_path):
    """The unchanged reference CLI, now running on the MI355X engine, against the WAV files the all-reference CLI
    wrote (tests/golden/wav_*.npz): header byte-identical, samples within one LSB of the truncating int16 writer."""
    if not _have():
        pytest.skip("oracle/_ref/audiomod-exe-mi355x not built")
    z = np.load(os.path.join(GOLD, f"wav_{name}.npz"))
    argv = ast.literal_eval(str(z["argv"]))
    fin, fout = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    open(fin, "wb").write(z["in_wav"].tobytes())
    r = subprocess.run([EXE, argv[0], fin, fout] + argv[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    got = np.frombuffer(open(fout, "rb").read(), np.uint8)
    want = z["out_wav"]
    assert got.size == want.size
    assert np.array_equal(got[:56], want[:56]), "WAV header differs"
    g = got[56:].view("<i2").astype(np.int32)
    w = want[56:].view("<i2").astype(np.int32)
    diff = np.abs(g - w)
    assert diff.max() <= 1
    assert (diff != 0).mean() < (0.25 if name == "constant" else 0.01)
    if name == "robotic":
        assert diff.max() == 0
