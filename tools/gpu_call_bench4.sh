cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python bench.py --steps 5 --warmup 2 > gpurun_out/b3_f32.json 2> gpurun_out/b3_f32.err
python bench.py --steps 5 --warmup 2 --int8 --no-cpu-baseline > gpurun_out/b3_i8.json 2> gpurun_out/b3_i8.err
python bench.py --steps 5 --warmup 2 --fast --no-cpu-baseline > gpurun_out/b3_f32_fast.json 2> gpurun_out/b3_f32_fast.err
python bench.py --steps 5 --warmup 2 --int8 --fast --no-cpu-baseline > gpurun_out/b3_i8_fast.json 2> gpurun_out/b3_i8_fast.err
for f in b3_f32 b3_i8 b3_f32_fast b3_i8_fast; do echo $f; tail -3 gpurun_out/$f.err; python -c "
import json,sys
d=json.load(open('gpurun_out/$f.json'))
print(d['value']/1e6, d['ms_per_step'], d.get('parity_checked'), d['roofline']['frac'], d['roofline']['launch_ms'], (d.get('cpu_baseline') or {}).get('value'))
"; done
