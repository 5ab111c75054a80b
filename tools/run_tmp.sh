one() { python bench.py --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -2 gpurun_out/ab.err; python -c "
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$1 |', round(d['value']/1e6,2), 'M parity', d.get('parity_checked'))"; }
for eh in 24 26 28 30; do LPCN_DEAL_EH=$eh one "EH=$eh"; done
for c in "5000,1600,900,200,170" "4000,1600,900,200,170" "4000,1000,600,150,130" "3000,1000,600,200,170"; do LPCN_DEAL_EH=28 LPCN_DEAL_COST=$c one "EH=28 COST=$c"; done
