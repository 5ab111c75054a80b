"""The generated constant tables (tools/gen_tables.py) against the reference's own tables, when the
reference tree is mounted; otherwise against their closed formulas only."""
import math
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_tables  # noqa: E402

REF = "/root/reference/src"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")


def _floats(text):
    return np.array([float(x) for x in re.findall(r"(-?\d+\.\d+(?:e-?\d+)?)f", text)], np.float32)


@needs_ref
def test_tansig_matches_reference_header():
    ref = _floats(open(os.path.join(REF, "tansig_table.h")).read())
    assert ref.size == 201 and np.array_equal(ref, gen_tables.tansig())


@needs_ref
def test_fft_and_dct_tables_match_reference():
    t = open(os.path.join(REF, "lpcnet_tables.c")).read()
    br = np.array([int(x) for x in re.findall(r"-?\d+", re.search(r"fft_bitrev\[\d*\]\s*=\s*\{([^}]*)\}", t).group(1))])
    assert np.array_equal(br, gen_tables.fft_bitrev())
    tw = _floats(re.search(r"fft_twiddles\[\d*\]\s*=\s*\{(.*?)\};", t, re.S).group(1)).reshape(-1, 2)
    assert np.array_equal(tw, gen_tables.fft_twiddles())
    dct = _floats(re.search(r"dct_table\[\d*\]\s*=\s*\{(.*?)\};", t, re.S).group(1))
    assert np.array_equal(dct, gen_tables.idct_table())


def test_headers_are_in_sync_with_generator():
    for path, prefix in ((os.path.join(ROOT, "lpcnet_amd", "csrc", "lpcnet_tables_gen.h"), "lpcn_"),
                         (os.path.join(ROOT, "oracle", "orc_tables_gen.h"), "orc_")):
        text = open(path).read()
        body = re.search(prefix + r"tansig\[201\] = \{(.*?)\};", text, re.S).group(1)
        vals = np.array([float.fromhex(x.rstrip("f")) for x in re.findall(r"-?0x[0-9a-fp.+-]+f", body)], np.float32)
        assert np.array_equal(vals, gen_tables.tansig())


def test_formula_sanity():
    u = gen_tables.ulaw2lin()
    assert u[128] == 0.0 and 31000 < u[255] < 32768 and np.all(np.diff(u) > 0)
    lg = gen_tables.logit_table()
    assert abs(lg[0] + math.log(0.975 / 0.025)) < 1e-5 and np.all(np.diff(lg) > 0)
    assert sorted(gen_tables.fft_bitrev().tolist()) == list(range(320))
