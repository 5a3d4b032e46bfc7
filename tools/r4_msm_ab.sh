#!/bin/bash
# MSM A/B on one box: default build, then GKRHIP_EXTRA_FLAGS="$1" rebuilt on the box.  Usage: bash tools/r4_msm_ab.sh "<flags>" [logn...]
FL=$1; shift
python tools/msm_bench.py ${@:-20 22 24}
GKRHIP_EXTRA_FLAGS="$FL" python -c "import importlib; importlib.import_module('gkr-mimc_amd.build').build(force=True)" > /dev/null 2>&1
export GKRHIP_EXTRA_FLAGS="$FL"
echo "--- $FL"
python tools/msm_bench.py ${@:-20 22 24}
