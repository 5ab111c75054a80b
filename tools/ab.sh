#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh libA.so libB.so ... (paths relative to the repo root); optional BENCH_ARGS
# prints M samples/s, streams per workgroup and the number of oracle-checked streams of each (two rounds, interleaved)
mkdir -p gpurun_out
for round in 1 2; do
  for lib in "$@"; do
    LPCNET_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 8 --warmup 2 $BENCH_ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
    python - "$lib" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
    print(sys.argv[1], '|', round(d['value'] / 1e6, 2), 'M | S =', d['config'].get('streams_per_workgroup'), '| parity_checked', d.get('parity_checked'), '| kernel ms', d['roofline'].get('kernel_ms'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
  done
done
