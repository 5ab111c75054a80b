#!/bin/bash
# kernel-rate A/B of experiment builds of the two-group kernel on one box: bash tools/x2_ab.sh <suffix> <suffix> ...   (libraries lpcnet_amd/liblpcnet_hip_<suffix>.so, two rounds, alternating)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export X2_SKIP_PARITY=1
for rep in 1 2; do for l in "$@"; do
  echo "$l: $(LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_$l.so timeout 100 python tests/tools/x2_check.py 6 2048 2>&1 | grep timing)"
done; done
