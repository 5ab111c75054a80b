tools/ab.sh lpcnet_amd/liblpcnet_hip_base.so lpcnet_amd/liblpcnet_hip.so 2>&1 | tee gpurun_out/ab3.log
for eh in 22 24; do echo EH=$eh; LPCN_DEAL_EH=$eh tools/ab.sh lpcnet_amd/liblpcnet_hip.so 2>&1 | head -1; done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
