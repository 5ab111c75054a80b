cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -k "int8" 2>&1 | tail -3
for spw in 1 2 4; do
python bench.py --steps 5 --warmup 2 --int8 --spw $spw --no-cpu-baseline > gpurun_out/b8_i8_s$spw.json 2> gpurun_out/b8.err
python -c "
import json,sys
d=json.load(open('gpurun_out/b8_i8_s$spw.json'))
print('int8 parity spw $spw', d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms'], d.get('parity_checked'))
"; done
python bench.py --steps 5 --warmup 2 --int8 --fast --no-cpu-baseline > gpurun_out/b8_i8_fast_auto.json 2> gpurun_out/b8.err
python -c "
import json,sys
d=json.load(open('gpurun_out/b8_i8_fast_auto.json'))
print('int8 fast auto', d['value']/1e6, d['ms_per_step'], d['config']['streams_per_workgroup'])
"
