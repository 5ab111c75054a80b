one() { python bench.py --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -2 gpurun_out/ab.err; python -c "
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$1 |', round(d['value']/1e6,2), 'M parity', d.get('parity_checked'))"; }
one "default"
for c in "5000,1600,900,395,320,500" "5000,1600,900,395,320,1000" "5000,1600,900,395,320,1500" "5000,1600,900,250,200,1000" "4500,1400,600,250,200,800"; do LPCN_DEAL_COST=$c one "COST=$c"; done
