cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_fast.py -x -q 2>&1 | tail -5
for spw in 2 4; do
python bench.py --steps 5 --warmup 2 --int8 --fast --spw $spw --no-cpu-baseline > gpurun_out/b5_i8_fast_s$spw.json 2> gpurun_out/b5.err
python -c "
import json,sys
d=json.load(open('gpurun_out/b5_i8_fast_s$spw.json'))
print('int8 fast spw $spw', d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms'])
"; done
