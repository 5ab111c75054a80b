timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "int8 or i8 or shapes" 2>&1 | tail -3
BENCH_ARGS="--int8" tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so
BENCH_ARGS="--int8 --spw 4" tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so | head -2
