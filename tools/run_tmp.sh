one() { LPCNET_HIP_LIB=$PWD/$1 python bench.py --no-cpu-baseline --steps 8 --warmup 2 $2 > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -2 gpurun_out/ab.err; python -c "
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$1 $2 |', round(d['value']/1e6,3), 'M parity', d.get('parity_checked'), 'S', d['config'].get('streams_per_workgroup'))"; }
for a in "--streams 1" "--streams 512 --spw 2" "--streams 256 --spw 1" ""; do for l in lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so; do one $l "$a"; done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
