import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: one HIP runtime per process, see tests/conftest.py)
from lpcnet_amd import api, synth
blob = synth.blob_bytes(synth.make_model())
step = sys.argv[1]
if step == "batch":
    b = api.LPCNetBatch(3, blob); f = np.stack([synth.make_features(1, 4)] * 3); b.synthesize(f)
elif step == "single":
    st = api.LPCNetState(blob); st.synthesize(synth.make_features(1, 2)[0])
elif step == "load":
    api.load_library()
import torch
print(step, "torch sees", torch.cuda.device_count(), torch.cuda.is_available())
x = torch.zeros(4).cuda(); print("ok")
