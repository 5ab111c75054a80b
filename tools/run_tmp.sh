tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip_prev.so lpcnet_amd/liblpcnet_hip.so
BENCH_ARGS="--int8" tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so | head -2
BENCH_ARGS="--fast" tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so | head -2
