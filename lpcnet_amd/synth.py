"""Seeded synthetic LPCNet model + feature generator.

The trained model (``lpcnet_data-<sha>.tar.gz``) is not in the reference tree and cannot be
downloaded, so every test / benchmark in this repo runs on weights produced here.  The arrays
are emitted with the *names, shapes and memory layouts* that the reference's exporter
``training_tf2/dump_lpcnet.py`` produces and that ``src/parse_lpcnet_weights.c`` binds, so the
same ``DNNw`` blob feeds the compiled reference (``oracle/_ref``), the C restatement
(``oracle/``) and the HIP engine.

Layout citations (all relative to /root/reference):
  * block-sparse recurrent matrix ........ training_tf2/dump_lpcnet.py:83-117 (printSparseVector)
  * float block = [in 4][out 8], int8 block = [out 8][in 4] ... dump_lpcnet.py:106-107
  * GRU-B recurrent int8 re-blocking ...... dump_lpcnet.py:58-59 (printVector dotp=True)
  * subias ................................ dump_lpcnet.py:131-133, 166-169
  * dual_fc transposes .................... dump_lpcnet.py:217-219
  * pre-multiplied embeddings ............. dump_lpcnet.py:333-339
  * block sparsifier (top-density 4x8 blocks + diagonal) ... training_tf2/lpcnet.py:96-116
  * blob record header .................... src/nnet.h:54-61, src/write_lpcnet_weights.c:47-66
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

N_A = 384          # GRU-A units          (lpcnet.py:234 rnn_units1)
N_B = 16           # GRU-B units          (lpcnet.py:234 rnn_units2)
COND = 128         # conditioning width   (lpcnet.py:234 cond_size)
EMBED = 128        # mu-law embedding     (lpcnet.py:47)
PITCH_EMBED = 64   # lpcnet.py:250
NB_FEATURES = 20   # include/lpcnet.h:45
NB_TOTAL_FEATURES = 36
FRAME_SIZE = 160
LPC_ORDER = 16
NB_BANDS = 18

WEIGHT_TYPE_FLOAT = 0
WEIGHT_TYPE_INT = 1
WEIGHT_TYPE_QWEIGHT = 2


@dataclass
class Model:
    """name -> (array, blob type).  ``flavour`` is 'float' or 'int8' (qweight element type)."""
    flavour: str
    arrays: dict = field(default_factory=dict)

    def add(self, name, arr, wtype):
        self.arrays[name] = (np.ascontiguousarray(arr), wtype)

    def get(self, name):
        return self.arrays[name][0]

    @property
    def nb_blocks_a(self):
        idx = self.get("sparse_gru_a_recurrent_weights_idx")
        return idx.size - 3 * N_A // 8

    @property
    def nb_blocks_b(self):
        idx = self.get("gru_b_weights_idx")
        return idx.size - 3 * N_B // 8


def _snap(w):
    """Snap to k/128 so that the float and the int8 flavours describe the same network."""
    return (np.round(w * 128.0) / 128.0).astype(np.float32)


def _pair_clip_q(q):
    """Integer-domain version of lpcnet.py:216-232 WeightClip: adjacent *input* pairs of a
    row must satisfy |q0|+|q1| <= 127 so a u8 x s8 pair-sum cannot saturate int16.
    q: int array [in, out] (in is even)."""
    q = q.copy()
    a = np.abs(q[0::2, :]) + np.abs(q[1::2, :])
    over = a > 127
    if over.any():
        scale = np.where(over, 127.0 / np.maximum(a, 1), 1.0)
        q0 = np.trunc(q[0::2, :] * scale).astype(q.dtype)
        q1 = np.trunc(q[1::2, :] * scale).astype(q.dtype)
        q[0::2, :] = q0
        q[1::2, :] = q1
    return q


def _quantize_matrix(w):
    """float [in,out] -> (float snapped to k/128, int q) with |q|<=127 and pair constraint."""
    q = np.clip(np.round(w * 128.0), -127, 127).astype(np.int64)
    q = _pair_clip_q(q)
    return (q / 128.0).astype(np.float32), q


def _block_sparsify(A, density, group_gain=None):
    """lpcnet.py:96-116 restated in the exporter's orientation: A is [in N][out N] for one gate;
    keep the top-`density` fraction of 4(in) x 8(out) blocks by energy (diagonal excluded).
    group_gain [N/8]: per output row group, a factor on the block energies the selection sees (the weights themselves are not scaled) --
    a stand-in for what TRAINED weights do to the sparsifier: some output units keep most of their inputs, others almost none."""
    N = A.shape[0]
    Ad = A - np.diag(np.diag(A))
    L = Ad.reshape(N // 4, 4, N // 8, 8)
    S = (L * L).sum(axis=3).sum(axis=1)                  # [N/4][N/8]
    if group_gain is not None:
        S = S * (np.asarray(group_gain, A.dtype)[None, :] ** 2)
    SS = np.sort(S.reshape(-1))
    thresh = SS[int(round(N * N // 32 * (1 - density)))]
    mask = (S >= thresh).astype(A.dtype)
    mask = np.repeat(np.repeat(mask, 4, axis=0), 8, axis=1)
    return Ad * mask                                      # diagonal handled separately


def _sparse_blocks(A, q):
    """dump_lpcnet.py:83-117: walk 8-wide output groups, 4-high input blocks.
    A float [in][out] (diag already removed), q its integer image.
    Returns (W0 float blocks [in4][out8], W int8 blocks [out8][in4], idx)."""
    n_in, n_out = A.shape
    W0, W, idx = [], [], []
    for i in range(n_out // 8):
        pos = len(idx)
        idx.append(-1)
        nb = 0
        for j in range(n_in // 4):
            block = A[j * 4:(j + 1) * 4, i * 8:(i + 1) * 8]
            if np.abs(block).sum() > 1e-10:
                nb += 1
                idx.append(j * 4)
                W0.append(block.reshape(-1))
                W.append(q[j * 4:(j + 1) * 4, i * 8:(i + 1) * 8].T.reshape(-1))
        idx[pos] = nb
    W0 = np.concatenate(W0).astype(np.float32) if W0 else np.zeros(0, np.float32)
    W = np.concatenate(W).astype(np.int8) if W else np.zeros(0, np.int8)
    return W0, W, np.asarray(idx, dtype=np.int32)


def make_model(seed=1234, flavour="float", shaped=True, densities=(0.05, 0.05, 0.2),
               lpc_gamma=1.0, grub_density=1.0, off_grid=False, skew=0.0):
    """Build the synthetic model.  The *same* seed gives the same network in both flavours
    (weights are snapped to k/128); only the qweight element type / blocking differs.
    off_grid=True (float flavour only) nudges the non-zero GRU weights off the k/128 grid, like a model trained
    without quantisation.
    skew > 0: TRAINED-LIKE block distribution of GRU-A -- the same number of blocks per gate (training_tf2/lpcnet.py:73-129 keeps the top-density
    fraction of a gate's blocks), but their count per 8-row group is heavy-tailed instead of near-uniform: the selection sees each row group's
    block energies through a log-normal gain exp(skew * N(0, 1)), so some candidate-gate groups keep 50-80 of their 96 possible blocks and others
    none (i.i.d. random weights give every group of a gate about the same count, which no trained model does; SURVEY section 7, hard part 3)."""
    assert flavour in ("float", "int8")
    rng = np.random.default_rng(seed)
    f32 = np.float32
    m = Model(flavour)

    def normal(shape, sigma):
        return (rng.standard_normal(shape) * sigma).astype(f32)

    # ---- mu-law embedding and GRU-A input kernel, folded (dump_lpcnet.py:333-339)
    E = normal((256, EMBED), 0.35)
    # smooth component so neighbouring mu-law levels look alike (PCMInit-like)
    ramp = (np.arange(256, dtype=f32)[:, None] - 127.5) / 127.5
    E += ramp * normal((1, EMBED), 0.5)
    Wa_in = normal((3 * EMBED + COND, 3 * N_A), 0.06)
    m.add("gru_a_embed_sig_weights", np.dot(E, Wa_in[0:EMBED]).astype(f32), WEIGHT_TYPE_FLOAT)
    m.add("gru_a_embed_pred_weights", np.dot(E, Wa_in[EMBED:2 * EMBED]).astype(f32), WEIGHT_TYPE_FLOAT)
    m.add("gru_a_embed_exc_weights", np.dot(E, Wa_in[2 * EMBED:3 * EMBED]).astype(f32), WEIGHT_TYPE_FLOAT)
    bias_a = normal((2, 3 * N_A), 0.1)
    m.add("gru_a_dense_feature_weights", Wa_in[3 * EMBED:], WEIGHT_TYPE_FLOAT)
    m.add("gru_a_dense_feature_bias", bias_a[0], WEIGHT_TYPE_FLOAT)

    # ---- GRU-B (input part 384->48 block-sparse at density 1, 128->48 dense, recurrent 16->48)
    Wb_in = normal((N_A + COND, 3 * N_B), 0.07)
    Wb_gru = Wb_in[:N_A]
    if grub_density < 1.0:
        # block-sparse GRU-B input matrix (lpcnet.py --grub-density-split): keep the strongest 4x8 blocks
        L = Wb_gru.reshape(N_A // 4, 4, 3 * N_B // 8, 8)
        energy = (L * L).sum(axis=3).sum(axis=1)
        thresh = np.sort(energy.reshape(-1))[int(round(energy.size * (1 - grub_density)))]
        Wb_gru = Wb_gru * np.repeat(np.repeat((energy >= thresh).astype(f32), 4, axis=0), 8, axis=1)
    Wb_a, Qb_a = _quantize_matrix(Wb_gru)
    m.add("gru_b_dense_feature_weights", Wb_in[N_A:], WEIGHT_TYPE_FLOAT)
    m.add("gru_b_dense_feature_bias", np.zeros(3 * N_B, f32), WEIGHT_TYPE_FLOAT)
    W0, W, idx = _sparse_blocks(Wb_a, Qb_a)
    if off_grid and flavour == "float":
        W0 = (W0 * (1.0 + 1e-3 * np.sin(np.arange(W0.size) * 0.37)).astype(f32)).astype(f32)
    m.add("gru_b_weights", W0 if flavour == "float" else W, WEIGHT_TYPE_QWEIGHT)
    m.add("gru_b_weights_idx", idx, WEIGHT_TYPE_INT)
    Wb_rec, Qb_rec = _quantize_matrix(normal((N_B, 3 * N_B), 0.25))
    if flavour == "float":
        m.add("gru_b_recurrent_weights", Wb_rec, WEIGHT_TYPE_QWEIGHT)           # [in 16][out 48]
    else:
        v = Qb_rec.reshape(N_B // 4, 4, 3 * N_B // 8, 8).transpose(2, 0, 3, 1)  # dump_lpcnet.py:58-59
        m.add("gru_b_recurrent_weights", v.astype(np.int8), WEIGHT_TYPE_QWEIGHT)
    bias_b = normal((2, 3 * N_B), 0.1)
    subias_b = bias_b.copy()
    subias_b[0] -= (Qb_a * (1.0 / 128.0)).sum(axis=0).astype(f32)
    subias_b[1] -= (Qb_rec * (1.0 / 128.0)).sum(axis=0).astype(f32)
    m.add("gru_b_bias", bias_b, WEIGHT_TYPE_FLOAT)
    m.add("gru_b_subias", subias_b.astype(f32), WEIGHT_TYPE_FLOAT)

    # ---- frame-rate network
    m.add("feature_conv1_weights", normal((3, NB_FEATURES + PITCH_EMBED, COND), 0.08), WEIGHT_TYPE_FLOAT)
    m.add("feature_conv1_bias", normal((COND,), 0.05), WEIGHT_TYPE_FLOAT)
    m.add("feature_conv2_weights", normal((3, COND, COND), 0.07), WEIGHT_TYPE_FLOAT)
    m.add("feature_conv2_bias", normal((COND,), 0.05), WEIGHT_TYPE_FLOAT)
    m.add("embed_pitch_weights", normal((256, PITCH_EMBED), 0.3), WEIGHT_TYPE_FLOAT)
    m.add("feature_dense1_weights", normal((COND, COND), 0.12), WEIGHT_TYPE_FLOAT)
    m.add("feature_dense1_bias", normal((COND,), 0.05), WEIGHT_TYPE_FLOAT)
    m.add("feature_dense2_weights", normal((COND, COND), 0.12), WEIGHT_TYPE_FLOAT)
    m.add("feature_dense2_bias", normal((COND,), 0.05), WEIGHT_TYPE_FLOAT)
    m.add("embed_sig_weights", E, WEIGHT_TYPE_FLOAT)      # bound by init, unused at inference

    # ---- dual FC (Keras shapes (256,16,2),(256,2),(256,2); dump_lpcnet.py:217-219 transposes)
    sig = 0.15 if shaped else 0.6
    Wd = normal((256, N_B, 2), sig)
    bd = normal((256, 2), 0.1 if shaped else 0.5)
    fd = (0.5 + np.abs(normal((256, 2), 0.5))).astype(f32)
    if shaped:
        # SURVEY §7 hard part 8: force tree bits 1..3 to the complement of the sign bit so the
        # excitation stays in mu-law [112,143] and the PCM is speech-like instead of saturating.
        for b in (1, 2, 3):
            for prefix in range(1 << b):
                i = (1 << b) | prefix
                first_bit = prefix >> (b - 1)
                bd[i, :] = -3.0 if first_bit else 3.0
                fd[i, :] = 2.5
                Wd[i] *= 0.2
    m.add("dual_fc_weights", np.transpose(Wd, (0, 2, 1)), WEIGHT_TYPE_FLOAT)    # [256][2][16]
    m.add("dual_fc_bias", np.transpose(bd, (1, 0)), WEIGHT_TYPE_FLOAT)           # [2][256]
    m.add("dual_fc_factor", np.transpose(fd, (1, 0)), WEIGHT_TYPE_FLOAT)         # [2][256]

    # ---- GRU-A recurrent: [in 384][out 1152], block-sparse per gate + diagonal
    A = normal((N_A, 3 * N_A), 1.0)
    gate_sigma = (0.30, 0.30, 0.22)
    diag = []
    skew_rng = np.random.default_rng(seed + 7919)         # (its own generator: the other weights of the model do not depend on `skew`)
    for k in range(3):
        Ak = A[:, k * N_A:(k + 1) * N_A] * gate_sigma[k]
        diag.append(_snap(np.diag(Ak) * 1.5))
        gain = np.exp(skew * skew_rng.standard_normal(N_A // 8)) if skew > 0 else None
        Ak = _block_sparsify(Ak, densities[k], gain)
        if skew > 0:
            # a row that keeps 60 blocks instead of 19 sums 3 x as many terms: scale the kept weights of a row group so that its pre-activation
            # keeps about the variance of the uniform model's rows (trained weights of well-connected units are smaller, too)
            cnt = (np.abs(Ak).reshape(N_A // 4, 4, N_A // 8, 8).sum(axis=(1, 3)) > 0).sum(axis=0)          # blocks per row group
            scale = np.sqrt(max(densities[k] * (N_A // 4), 1.0) / np.maximum(cnt, 1.0))
            Ak = Ak * np.repeat(np.minimum(scale, 1.5), 8)[None, :].astype(f32)
        A[:, k * N_A:(k + 1) * N_A] = Ak
    A, QA = _quantize_matrix(A)
    W0, W, idx = _sparse_blocks(A, QA)
    m.add("sparse_gru_a_recurrent_weights_diag", np.concatenate(diag), WEIGHT_TYPE_FLOAT)
    if off_grid and flavour == "float":
        W0 = (W0 * (1.0 + 1e-3 * np.cos(np.arange(W0.size) * 0.23)).astype(f32)).astype(f32)
    m.add("sparse_gru_a_recurrent_weights", W0 if flavour == "float" else W, WEIGHT_TYPE_QWEIGHT)
    m.add("sparse_gru_a_recurrent_weights_idx", idx, WEIGHT_TYPE_INT)
    subias_a = bias_a.copy()
    subias_a[1] -= (QA * (1.0 / 128.0)).sum(axis=0).astype(f32)
    m.add("sparse_gru_a_bias", bias_a, WEIGHT_TYPE_FLOAT)
    m.add("sparse_gru_a_subias", subias_a.astype(f32), WEIGHT_TYPE_FLOAT)
    m.lpc_gamma = lpc_gamma
    return m


def blob_bytes(model: Model) -> bytes:
    """Serialise as the reference's ``DNNw`` weight blob (src/nnet.h:54-61,
    src/write_lpcnet_weights.c:47-66): 64-byte header + payload padded to 64."""
    out = bytearray()
    for name, (arr, wtype) in model.arrays.items():
        raw = arr.tobytes()
        size = len(raw)
        block = (size + 63) // 64 * 64
        nm = name.encode()
        assert len(nm) < 44
        out += struct.pack("<4siiii44s", b"DNNw", 0, wtype, size, block, nm)
        out += raw + b"\0" * (block - size)
    return bytes(out)


def write_blob(model: Model, path):
    with open(path, "wb") as f:
        f.write(blob_bytes(model))


def make_features(stream_seed: int, n_frames: int) -> np.ndarray:
    """(n_frames, 36) float32 in ``lpcnet_demo -synthesis`` input format; only [0..19] are read
    (src/lpcnet_demo.c:213-215).  SURVEY §8(d): c0 in [-8,2], c1..17 ~ N(0,1)+slow sinusoid,
    f[18]=(P-200)/100 with P in [66,510] slowly varying, f[19] in [-0.5,0.5]."""
    rng = np.random.default_rng(stream_seed)
    t = np.arange(n_frames, dtype=np.float64)
    f = np.zeros((n_frames, NB_TOTAL_FEATURES), np.float64)
    ph = rng.uniform(0, 2 * np.pi, size=NB_BANDS)
    rate = rng.uniform(0.01, 0.08, size=NB_BANDS)
    f[:, 0] = -3.0 + 5.0 * np.sin(rate[0] * t + ph[0])
    scale = 1.0 / (1.0 + 0.25 * np.arange(1, NB_BANDS))
    f[:, 1:NB_BANDS] = (0.35 * rng.standard_normal((n_frames, NB_BANDS - 1))
                        + np.sin(rate[1:] * t[:, None] + ph[1:])) * scale
    P = 288.0 + 222.0 * np.sin(rng.uniform(0.005, 0.03) * t + rng.uniform(0, 2 * np.pi))
    f[:, NB_BANDS] = (P - 200.0) / 100.0
    f[:, NB_BANDS + 1] = 0.5 * np.sin(rng.uniform(0.01, 0.05) * t + rng.uniform(0, 2 * np.pi))
    return f.astype(np.float32)


def make_codebooks(seed: int = 5):
    """Synthetic VQ codebooks for the codec path (the reference's ceps_codebooks.c is a generated
    file that is not in its tree): cb1..3 [1024][17], diff4 [4096][18]; src/lpcnet_dec.c:130-141."""
    rng = np.random.default_rng(seed)
    cb1 = (rng.standard_normal((1024, 17)) * 1.2).astype(np.float32)
    cb2 = (rng.standard_normal((1024, 17)) * 0.5).astype(np.float32)
    cb3 = (rng.standard_normal((1024, 17)) * 0.25).astype(np.float32)
    cbd = (rng.standard_normal((4096, 18)) * 0.3).astype(np.float32)
    return cb1, cb2, cb3, cbd
