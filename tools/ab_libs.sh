#!/bin/bash
# A/B of experiment builds (LPCN_LIB_SUFFIX libraries) on one box: bench.py's headline figure, alternating the libraries, three rounds
#   bash tools/ab_libs.sh "<bench flags>" lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
flags=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    v=$(LPCNET_HIP_NO_AUTOTUNE=1 LPCNET_HIP_LIB=$PWD/lpcnet_amd/$lib timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $flags 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f M  launch %.3f ms  S=%s parity=%s' % (d['value']/1e6, d['roofline']['launch_ms'], d['config']['streams_per_workgroup'], d['parity_checked']))" 2>&1 | tail -1)
    echo "$lib [$flags]: $v"
  done
done
