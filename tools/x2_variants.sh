#!/bin/bash
# timing of experiment builds of the two-group kernel (LPCN_LIB_SUFFIX builds): X2_SKIP_PARITY=1 python tests/tools/x2_check.py per library
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export X2_SKIP_PARITY=1
for lib in lpcnet_amd/liblpcnet_hip.so $(ls lpcnet_amd/liblpcnet_hip_*.so | grep -v "prof\|smallreg"); do
  for rep in 1 2; do echo "== $lib"; LPCNET_HIP_LIB=$PWD/$lib timeout 100 python tests/tools/x2_check.py 6 2048 2>&1 | grep timing; done
done
