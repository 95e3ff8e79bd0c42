import ast
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def e2e_cases():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLD, "e2e_*.npz")))


def load_e2e(name):
    z = np.load(os.path.join(GOLD, f"e2e_{name}.npz"))
    x = z["x_i16"].astype(np.float32) / np.float32(32768.0)
    meta = ast.literal_eval(str(z["meta"]))
    return x, z["y"], [int(v) for v in z["counts"]], meta


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))
