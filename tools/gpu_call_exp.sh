cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b6_f32.json 2> gpurun_out/b6.err
python bench.py --steps 5 --warmup 2 --int8 --no-cpu-baseline > gpurun_out/b6_i8.json 2>> gpurun_out/b6.err
python bench.py --steps 5 --warmup 2 --int8 --fast --spw 2 --no-cpu-baseline > gpurun_out/b6_i8_fast_s2.json 2>> gpurun_out/b6.err
python bench.py --steps 5 --warmup 2 --fast --no-cpu-baseline > gpurun_out/b6_f32_fast.json 2>> gpurun_out/b6.err
for f in b6_f32 b6_i8 b6_i8_fast_s2 b6_f32_fast; do python -c "
import json,sys
d=json.load(open('gpurun_out/$f.json'))
print('$f', d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms'], d.get('parity_checked'))
"; done
LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/gpu_sweep.py 22 1024:4 2>&1 | tail -10
