#!/usr/bin/env python3
"""The arithmetic identities the bit-exact (PARITY) kernels rest on, measured on the device (needs a GPU).

    python tools/arith_identities.py [--lanes-per-stratum N] > profiles/rNN_arith_identities.json

The float PARITY kernels form GRU-A's products with v_mfma_f32_4x4x1 and C = -0.0 (an exact multiplier: the fused result
of a*b + (-0) is the once-rounded product) and sum / multiply in pairs with v_pk_add_f32 / v_pk_mul_f32.  This asks the
device -- through the product library's own test seam, i.e. with its compile flags and float mode -- for the bit patterns
of both forms over operand pairs STRATIFIED to where a matrix pipe or a packed unit could differ from the vector unit:
subnormal inputs, products that land in the subnormal range or underflow, +-0, the largest finite products, overflow,
infinities.  Also compares the vector unit's product with IEEE-754 binary32 multiplication as numpy computes it on the
host (x86 SSE, denormals preserved).  tests/test_gpu_parity.py runs the same function and asserts zero mismatches.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _f32(sign, exp, man):
    return ((sign.astype(np.uint32) << 31) | (exp.astype(np.uint32) << 23) | man.astype(np.uint32)).view(np.float32)


def strata(m, seed=5):
    """name -> (a, b): m operand pairs each (m a multiple of 64; quads of lanes share a stratum, the matrix pipe mixes a quad's lanes)"""
    rng = np.random.default_rng(seed)
    sgn = lambda: rng.integers(0, 2, m)
    man = lambda: rng.integers(0, 1 << 23, m)
    ex = lambda lo, hi: rng.integers(lo, hi + 1, m)                  # biased exponents, inclusive
    sub = lambda: _f32(sgn(), np.zeros(m, np.int64), rng.integers(1, 1 << 23, m))
    out = {}
    out["normal x normal, random bit patterns (all exponents)"] = (_f32(sgn(), ex(1, 254), man()), _f32(sgn(), ex(1, 254), man()))
    out["well scaled: state in (-1, 1) x weight on the k/128 grid"] = (
        rng.uniform(-1, 1, m).astype(np.float32), (rng.integers(-127, 128, m) / 128.0).astype(np.float32))
    out["subnormal state x normal weight in [1/128, 8)"] = (sub(), _f32(sgn(), ex(120, 129), man()))
    out["normal x subnormal"] = (_f32(sgn(), ex(100, 150), man()), sub())
    out["subnormal x subnormal (underflow to zero)"] = (sub(), sub())
    ea = ex(30, 90)
    out["normal x normal, product in the subnormal range"] = (_f32(sgn(), ea, man()), _f32(sgn(), np.clip(127 - 149 - (ea - 127) + rng.integers(0, 24, m), 1, 254), man()))
    out["normal x normal, product at the normal / subnormal boundary"] = (_f32(sgn(), ea, man()), _f32(sgn(), np.clip(127 - 126 - (ea - 127) + rng.integers(-2, 2, m), 1, 254), man()))
    z = _f32(sgn(), np.zeros(m, np.int64), np.zeros(m, np.int64))
    mix = np.where(rng.integers(0, 4, m) == 0, _f32(sgn(), np.zeros(m, np.int64), np.zeros(m, np.int64)), _f32(sgn(), ex(1, 254), man()))
    out["+-0 x finite (and +-0 x +-0)"] = (z, mix.astype(np.float32))
    eb = ex(128, 250)
    out["largest finite products and overflow"] = (_f32(sgn(), eb, man()), _f32(sgn(), np.clip(127 + 127 - (eb - 127) + rng.integers(-2, 2, m), 1, 254), man()))
    inf = _f32(sgn(), np.full(m, 255), np.zeros(m, np.int64))
    out["infinity x finite non-zero"] = (inf, _f32(sgn(), ex(1, 254), man()))
    consts = np.array([np.finfo(np.float32).max, np.finfo(np.float32).tiny, 1.0, -1.0, 1e-45, -1e-45, 0.5, 2.0, 1.17549421e-38, 3.0e-39,
                       -3.0e-39, 16777216.0, 1.0 / 128, 127.0 / 128, 0.0, -0.0], np.float32)
    out["constants (FLT_MAX, FLT_MIN, smallest subnormal, 1, ...) x constants"] = (rng.choice(consts, m), rng.choice(consts, m))
    return out


def run(lanes_per_stratum=262144, seed=5):
    from lpcnet_amd import api
    m = (lanes_per_stratum + 63) // 64 * 64
    st = strata(m, seed)
    a = np.concatenate([v[0] for v in st.values()]).astype(np.float32)
    b = np.concatenate([v[1] for v in st.values()]).astype(np.float32)
    mf, mu, pk, sc = api.arith_identities(a, b)
    with np.errstate(all="ignore"):
        aq = a.reshape(-1, 4)                                          # product k of lane i = a[4 (i/4) + k] * b[i]
        want = (np.repeat(aq, 4, axis=0) * b[:, None]).astype(np.float32).view(np.uint32)
    nan = lambda u: (u & 0x7FFFFFFF) > 0x7F800000
    rep = {"library": api.build_info(), "lanes_per_stratum": m, "strata": {}}
    tot = dict(products=0, mfma_vs_mul=0, mul_vs_ieee=0, pk_vs_scalar=0)
    for k, name in enumerate(st):
        sl = slice(k * m, (k + 1) * m)
        d1 = (mf[sl] != mu[sl]) & ~(nan(mf[sl]) & nan(mu[sl]))
        d2 = (mu[sl] != want[sl]) & ~(nan(mu[sl]) & nan(want[sl]))
        d3 = (pk[sl] != sc[sl]) & ~(nan(pk[sl]) & nan(sc[sl]))
        res = mu[sl]
        rec = {"products": int(d1.size), "mfma_c_negzero_vs_v_mul_f32_mismatches": int(d1.sum()), "v_mul_f32_vs_ieee_host_mismatches": int(d2.sum()),
               "packed_halves": int(d3.size), "v_pk_mul_add_vs_scalar_mismatches": int(d3.sum()),
               "results_subnormal": int((((res & 0x7F800000) == 0) & ((res & 0x7FFFFF) != 0)).sum()), "results_zero": int(((res & 0x7FFFFFFF) == 0).sum()),
               "results_inf": int(((res & 0x7FFFFFFF) == 0x7F800000).sum())}
        if d1.any():
            i, kk = np.argwhere(d1)[0]
            rec["first_mismatch"] = {"a": float(a[sl][(i // 4) * 4 + kk]), "b": float(b[sl][i]), "mfma": hex(int(mf[sl][i, kk])), "v_mul": hex(int(mu[sl][i, kk]))}
        rep["strata"][name] = rec
        tot["products"] += int(d1.size); tot["mfma_vs_mul"] += int(d1.sum()); tot["mul_vs_ieee"] += int(d2.sum()); tot["pk_vs_scalar"] += int(d3.sum())
    rep["total"] = tot
    return rep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes-per-stratum", type=int, default=262144)
    a = ap.parse_args()
    print(json.dumps(run(a.lanes_per_stratum), indent=1))
