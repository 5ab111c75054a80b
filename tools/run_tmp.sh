timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
BENCH_ARGS="--streams 512 --spw 2" tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so
