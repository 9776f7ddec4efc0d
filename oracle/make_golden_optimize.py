#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: writes tests/golden/optimize_h_proposals.json — the first proposals of the reference's check-matrix
local search (optimize_H.cpp:66-75 under std::mt19937(239), optimize_H.cpp:132), produced by the REAL reference compiled as it
lies (oracle/ref_optimize_shim.cpp -> oracle/_ref/libacg_ref_opt.so, built by `make -C oracle ref` in the build container).
The fixture is data: (block row, block column, present, shift) per proposal; tests/test_drivers.py holds tools/drivers/
acg_optimize_h --dump-proposals to it.  Acceptance is not pinned (it depends on a 200-thread seed race): both
extreme chains are — every proposal rejected (all proposals mutate the start matrix) and every proposal accepted."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def proposals(lib, path, Z, seed, count, accept):
    out = np.zeros((count, 4), dtype=np.int32)
    rc = lib.ref_opt_proposals(path.encode(), Z, seed, count, accept, out.ctypes.data_as(C.c_void_p))
    assert rc == 0, "the reference's random_permute does not behave as the shim expects (rc %d)" % rc
    return out.tolist()


def main():
    lib = C.CDLL(os.path.join(HERE, "_ref", "libacg_ref_opt.so"))
    lib.ref_opt_proposals.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_void_p]
    fix = {"source": "optimize_H.cpp:66-75 (PermutationsMatrix::random_permute) driven as optimize_H.cpp:89-104 with std::mt19937(seed)",
           "columns": ["block_row", "block_col", "present", "shift"], "cases": []}
    for name, Z in (("H05.txt", 20), ("optimalH.txt", 20)):
        for seed in (239, 7):
            for accept in (0, 1):
                fix["cases"].append({"matrix": name, "Z": Z, "seed": seed, "accept_all": accept,
                                     "proposals": proposals(lib, os.path.join(ROOT, "data", name), Z, seed, 64, accept)})
    with open(os.path.join(ROOT, "tests", "golden", "optimize_h_proposals.json"), "w") as f:
        json.dump(fix, f)
    print("wrote %d cases" % len(fix["cases"]))


if __name__ == "__main__":
    main()
