"""Golden vectors of computeH (prover/gadget/prove.go:308-359) from the Python restatement oracle/pyoracle_fft.py
(gnark-crypto's fft.Domain is not under /root/reference: the restatement follows its published algorithm and is checked
against schoolbook polynomial division in tests/test_oracle.py -- "parity unpinned").  Run: python tests/golden/gen_compute_h.py"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle_fft as F  # noqa: E402

random.seed(20260303)
out = []
for n, card in ((1, 0), (2, 0), (3, 0), (5, 8), (8, 0), (13, 16), (16, 0), (37, 64), (64, 128)):
    a = [random.randrange(F.Q) for _ in range(n)]
    b = [random.randrange(F.Q) for _ in range(n)]
    # satisfied constraints on the first half (c = a*b), arbitrary values on the rest: both regimes of the formula
    c = [(x * y) % F.Q if i < n // 2 + 1 else random.randrange(F.Q) for i, (x, y) in enumerate(zip(a, b))]
    h = F.compute_h(a, b, c, card or None)
    out.append({"n": n, "cardinality": card, "a": [hex(v) for v in a], "b": [hex(v) for v in b], "c": [hex(v) for v in c],
                "h": [hex(v) for v in h]})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "compute_h.json"), "w"), indent=0)
print("wrote", len(out), "cases")
