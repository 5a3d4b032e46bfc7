"""Kernel launches of one gkr.Prove: python tools/launch_count.py [bn] [proofs] -- run under
rocprofv3 --kernel-trace --stats --output-format csv and divide the Calls column's sum (minus the setup) by the number of proofs;
prints its own marker lines so that the setup can be told apart."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gk = importlib.import_module("gkr-mimc_amd")
gk.init(0)
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = gk.MimcSession(bn)
s.synth_inputs()
s.assign()
q = np.arange(1, 4 * bn + 1, dtype=np.uint64).reshape(bn, 4)
for _ in range(proofs):
    s.prove(q)
print("proved %d times at bN = %d" % (proofs, bn))
