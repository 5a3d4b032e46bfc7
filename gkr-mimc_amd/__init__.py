"""gkr-mimc_amd -- MI355X-native GKR/sumcheck prover for batched MiMC7/BN254 (hot path of
Consensys/gkr-mimc).  The package holds the HIP sources (csrc/), the built C-ABI library
(libgkrhip.so) and a thin Python mirror of the reference's Go API for that path (prover.py), used
by the tests and the benchmark.  The directory name contains a hyphen: import it with

    import importlib; gk = importlib.import_module("gkr-mimc_amd")
"""
from .prover import *  # noqa: F401,F403
from . import prover  # noqa: F401
