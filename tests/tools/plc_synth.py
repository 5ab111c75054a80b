"""TEST TOOLING: a seeded synthetic PLC network (src/lpcnet_plc.c:135-145: dense 57 -> 128 tanh, GRU 128 -> 16, GRU 16 -> 16,
dense 16 -> 20 linear) in the array names / layouts training_tf2/dump_plc.py emits and oracle/refgen.py's init_plc_model binds,
appended to an LPCNet blob so that lpcnet_plc_load_model() (src/lpcnet_plc.c:88-97) finds both networks in one DNNw blob.
The packet-loss concealment itself is outside the HIP engine's scope: this exists to RUN the reference's unmodified
lpcnet_plc.c on top of the engine and compare it with the reference's own build (tests/test_demo_integration.py)."""
import numpy as np

from lpcnet_amd import synth

PLC_IN, PLC_D1, PLC_G1, PLC_G2, NB_FEATURES = 2 * 18 + 20 + 1, 128, 16, 16, 20


def _gru(m, name, rng, n_in, n, flavour):
    f32 = np.float32
    W, Q = synth._quantize_matrix((rng.standard_normal((n_in, 3 * n)) * 0.08).astype(f32))
    W0, Wq, idx = synth._sparse_blocks(W, Q)
    m.add(name + "_weights", W0 if flavour == "float" else Wq, synth.WEIGHT_TYPE_QWEIGHT)
    m.add(name + "_weights_idx", idx, synth.WEIGHT_TYPE_INT)
    Wr, Qr = synth._quantize_matrix((rng.standard_normal((n, 3 * n)) * 0.2).astype(f32))
    if flavour == "float":
        m.add(name + "_recurrent_weights", Wr, synth.WEIGHT_TYPE_QWEIGHT)
    else:
        m.add(name + "_recurrent_weights", Qr.reshape(n // 4, 4, 3 * n // 8, 8).transpose(2, 0, 3, 1).astype(np.int8), synth.WEIGHT_TYPE_QWEIGHT)
    bias = (rng.standard_normal((2, 3 * n)) * 0.1).astype(f32)
    sub = bias.copy()
    sub[0] -= (Q * (1.0 / 128.0)).sum(axis=0).astype(f32)
    sub[1] -= (Qr * (1.0 / 128.0)).sum(axis=0).astype(f32)
    m.add(name + "_bias", bias, synth.WEIGHT_TYPE_FLOAT)
    m.add(name + "_subias", sub.astype(f32), synth.WEIGHT_TYPE_FLOAT)


def make_model_with_plc(seed=4321, flavour="float", **kw):
    m = synth.make_model(flavour=flavour, **kw)
    rng = np.random.default_rng(seed)
    f32 = np.float32
    m.add("plc_dense1_weights", (rng.standard_normal((PLC_IN, PLC_D1)) * 0.1).astype(f32), synth.WEIGHT_TYPE_FLOAT)
    m.add("plc_dense1_bias", (rng.standard_normal(PLC_D1) * 0.05).astype(f32), synth.WEIGHT_TYPE_FLOAT)
    _gru(m, "plc_gru1", rng, PLC_D1, PLC_G1, flavour)
    _gru(m, "plc_gru2", rng, PLC_G1, PLC_G2, flavour)
    # the output layer predicts the 20 features: keep them in the range real features live in (c0 offset, small pitch/corr)
    m.add("plc_out_weights", (rng.standard_normal((PLC_G2, NB_FEATURES)) * 0.3).astype(f32), synth.WEIGHT_TYPE_FLOAT)
    bias = (rng.standard_normal(NB_FEATURES) * 0.2).astype(f32)
    bias[0] -= 3.0
    m.add("plc_out_bias", bias, synth.WEIGHT_TYPE_FLOAT)
    return m
