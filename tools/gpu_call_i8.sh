cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_wide.py -x -q -k "two_workgroups" 2>&1 | tail -4
for fl in "" "--fast"; do for cfg in "1024 0" "2048 0" "2048 4" "4096 4" "4096 2" "8192 4"; do set -- $cfg
python bench.py --steps 3 --warmup 1 --int8 $fl --streams $1 --spw $2 --no-cpu-baseline --check-streams 0 > gpurun_out/b9.json 2> gpurun_out/b9.err
python -c "
import json,sys
d=json.load(open('gpurun_out/b9.json'))
print('int8 $fl streams $1 spw $2 -> S=%d: %.1f M  step %.2f ms' % (d['config']['streams_per_workgroup'], d['value']/1e6, d['ms_per_step']))
"; done; done
