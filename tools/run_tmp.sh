BENCH_ARGS="--int8" tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so
BENCH_ARGS="--int8 --fast --spw 2" tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so | head -2
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "int8 or i8" 2>&1 | tail -3
