cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/sq
python tools/profile_sq.py --tag r02a > gpurun_out/sq/sq_f32.log 2>&1
python tools/profile_sq.py --int8 --tag r02a > gpurun_out/sq/sq_i8.log 2>&1
LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/gpu_sweep.py 22 1024:4 > gpurun_out/sq/phase_f32.log 2>&1
LPCN_FLAVOUR=int8 LPCNET_HIP_LIB=$PWD/lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/gpu_sweep.py 22 1024:4 > gpurun_out/sq/phase_i8.log 2>&1
tail -30 gpurun_out/sq/sq_f32.log; tail -14 gpurun_out/sq/phase_f32.log
