"""helpers shared by the oracle (CPU) and HIP (GPU) parity tests"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MATS = ["H", "H05", "optimalH"]
SNRS = [-3.0, -2.0, 0.0, 2.0]
BP_ITERS = [1, 2, 5, 20, 50]
ADMM_ITERS = [1, 2, 5, 10, 100]


def load(name, snr):
    z = np.load(os.path.join(GOLDEN, "%s_snr%+.0f.npz" % (name, snr)))
    return {k: z[k] for k in z.files}


def unpack(packed, n):
    return np.unpackbits(packed, axis=1)[:, :n]


def known():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)
