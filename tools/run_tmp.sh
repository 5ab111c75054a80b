for d in "0.06,0.06,0.22" "0.07,0.07,0.25" "0.08,0.08,0.3"; do
  python bench.py --no-cpu-baseline --steps 6 --warmup 2 --densities $d > gpurun_out/dens_$d.json 2> gpurun_out/ab.err || tail -2 gpurun_out/ab.err
  python -c "
import json
d=json.loads(open('gpurun_out/dens_$d.json').read().strip().splitlines()[-1]); print('$d |', round(d['value']/1e6,2), 'M parity', d.get('parity_checked'), 'S', d['config']['streams_per_workgroup'], 'blocks', d['config']['gru_a_blocks'], 'frac', round(d['roofline']['frac'],3))"
done
