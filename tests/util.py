"""Shared helpers for the tests: fixture loading and hex <-> fr.Element-image conversion."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def hex_to_fr(hs):
    """list of 64-hex strings (limb0 first) -> (n,4) uint64 Montgomery array."""
    if isinstance(hs, str):
        hs = [hs]
    out = np.zeros((len(hs), 4), dtype=np.uint64)
    for i, s in enumerate(hs):
        for k in range(4):
            out[i, k] = int(s[16 * k:16 * k + 16], 16)
    return out


def fr_to_hex(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return ["".join("%016x" % int(v) for v in row) for row in arr]
