"""CPU: the constant tables of the lookahead path, in the oracle and in the shipped sources, against the reference's own
data (tests/golden/reference_tables.json, read out of the reference's sources by tests/golden/make_reference_tables.py):
search patterns, the lookahead lambda per bit depth, x265_exp2_lut, the intra filter flags, the angle tables, the 8-tap
luma filter and the temporal filter's interpolation taps.  The device code keeps several of these packed (biased nibbles, one byte per
entry, a chain of selects); they are unpacked here from the text of the sources, the way the device unpacks them."""
import json
import os
import re

import numpy as np

from x265_amd import api
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))


def _text(*path):
    return open(os.path.join(ROOT, *path)).read()


def _initialiser(text, name):
    """Numbers of the brace initialiser that follows `name`, comments dropped."""
    at = text.index(name)
    a = text.index("=", at)
    b = text.index("};", a)
    body = re.sub(r"//[^\n]*|/\*.*?\*/", "", text[a + 1:b], flags=re.S)
    return [int(v, 0) for v in re.findall(r"-?(?:0x[0-9a-fA-F]+|\d+)", body)]


def _pairs(flat):
    return [list(p) for p in zip(flat[0::2], flat[1::2])]


def _constant(text, name):
    return int(re.search(r"\b%s\s*=\s*(0x[0-9A-Fa-f]+)u" % name, text).group(1), 16)


def _nibbles(pack, n=8):
    # nib() of x265_amd/csrc/device_common.h: entry i in bits 4i..4i+3, biased by 8
    return [((pack >> (4 * i)) & 15) - 8 for i in range(n)]


def test_fixture_is_whole():
    assert len(GOLDEN["hex2"]) == 8 and len(GOLDEN["square1"]) == 9 and len(GOLDEN["hex4"]) == 16
    assert len(GOLDEN["exp2_lut"]) == 64 and len(GOLDEN["intra_filter_flags"]) == 35
    assert len(GOLDEN["interpolation_filter"]) == 16 and all(sum(r) == 64 for r in GOLDEN["interpolation_filter"])


def test_oracle_search_patterns():
    src = _text("oracle", "orc_motion.c")
    assert _pairs(_initialiser(src, "k_hex2[8]")) == GOLDEN["hex2"]
    assert _initialiser(src, "k_mod6m1[8]") == GOLDEN["mod6m1"]
    assert _pairs(_initialiser(src, "k_square1[9]")) == GOLDEN["square1"]
    assert _pairs(_initialiser(src, "k_hex4[16]")) == GOLDEN["hex4"]


def test_device_search_patterns():
    src = _text("x265_amd", "csrc", "search_kernels.hip")
    hex2 = GOLDEN["hex2"]
    assert _nibbles(_constant(src, "HEX2_X")) == [p[0] for p in hex2]
    assert _nibbles(_constant(src, "HEX2_Y")) == [p[1] for p in hex2]
    # the six distinct points in the order hex2[1..6] visits them
    assert _nibbles(_constant(src, "HEX6_X"), 6) == [p[0] for p in hex2[1:7]]
    assert _nibbles(_constant(src, "HEX6_Y"), 6) == [p[1] for p in hex2[1:7]]
    sq = GOLDEN["square1"]
    assert sq[0] == [0, 0]
    assert _nibbles(_constant(src, "SQ1_X")) == [p[0] for p in sq[1:]]
    assert _nibbles(_constant(src, "SQ1_Y")) == [p[1] for p in sq[1:]]
    mod = _constant(src, "MOD6M1")
    assert [(mod >> (4 * i)) & 15 for i in range(8)] == GOLDEN["mod6m1"]
    h4 = GOLDEN["hex4"]
    assert _nibbles(_constant(src, "H4X0")) + _nibbles(_constant(src, "H4X1")) == [p[0] for p in h4]
    assert _nibbles(_constant(src, "H4Y0")) + _nibbles(_constant(src, "H4Y1")) == [p[1] for p in h4]


def test_lookahead_lambda(oracle_built):
    for depth in (8, 10, 12):
        want = GOLDEN["lookahead_lambda"][str(depth)]
        assert GOLDEN["lookahead_qp"][str(depth)] == 12 + 6 * (depth - 8)
        assert api.LOOKAHEAD_LAMBDA[depth] == want
        assert orc.lib(depth).orc_lookahead_lambda() == want


def test_exp2_lut(oracle_built):
    lut = GOLDEN["exp2_lut"]
    # the closed form both the oracle and the host driver fill their tables from
    assert [int(np.floor(256.0 * (2.0 ** (k / 64.0) - 1.0) + 0.5)) for k in range(64)] == lut
    # and the oracle's function at the 64 arguments that land on index 512 + k, where it returns lut[k] + 256
    L = orc.lib(8)
    assert [L.orc_exp2fix8(-k * 6.0 / 64.0) - 256 for k in range(64)] == lut
    for path in (("oracle", "orc_lookahead.c"), ("x265_amd", "host", "lookahead_host.cpp")):
        assert re.search(r"lround\(256\.0 \* \((?:pow\(2\.0, |exp2\()(?:i|k) / 64\.0\) - 1\.0\)\)", _text(*path)), path


def test_intra_filter_flags(oracle_built):
    flags = GOLDEN["intra_filter_flags"]
    assert _initialiser(_text("oracle", "orc_intra.c"), "k_filter_flags[35]") == flags
    L = orc.lib(8)
    for size in (8, 16, 32):
        assert [L.orc_intra_filter_flag(m, size) for m in range(35)] == [int(bool(f & size)) for f in flags]
    # the device's 8x8 test names the modes outright
    dev = re.search(r"bool uses_filtered\(int mode\) \{ return ([^;]*); \}", _text("x265_amd", "csrc", "intra_kernels.hip")).group(1)
    assert sorted(int(v) for v in re.findall(r"mode == (\d+)", dev)) == [m for m in range(35) if flags[m] & 8]


def test_angle_tables():
    angles, inv = GOLDEN["angle_table"], GOLDEN["inv_angle_table"]
    src = _text("oracle", "orc_intra.c")
    assert _initialiser(src, "angles[17]") == angles
    assert _initialiser(src, "invAngles[8]") == inv
    assert angles[8] == 0 and [-a for a in angles[:8]] == angles[:8:-1]
    dev = _text("x265_amd", "csrc", "intra_kernels.hip")
    # |angle| for offsets 1..8, one byte each, low byte first
    mag = int(re.search(r"\(0x([0-9A-Fa-f]{16})ull >> \(8 \* \(\(aabs - 1\) & 7\)\)\)", dev).group(1), 16)
    assert [(mag >> (8 * k)) & 0xff for k in range(8)] == angles[9:]
    # the inverse angle for offsets -1..-8, as a tree of selects on |offset|
    expr = re.search(r"const int inv = ([^;]*);", dev).group(1)

    def select(text, aabs):
        # conditions first, then the innermost "cond ? a : b" until one number is left
        py = re.sub(r"aabs (<=|==) (\d+)", lambda m: str(aabs <= int(m.group(2)) if m.group(1) == "<=" else aabs == int(m.group(2))),
                    re.sub(r"\s+", " ", text))
        py = re.sub(r"\((True|False)\)", r"\1", py)
        while "?" in py:
            py, n = re.subn(r"\(?(True|False) \? (\d+) : (\d+)\)?", lambda m: m.group(2) if m.group(1) == "True" else m.group(3), py, count=1)
            assert n == 1, py
        return int(py)

    assert [select(expr, k) for k in range(1, 9)] == inv


def test_temporal_filter_taps():
    taps = GOLDEN["interpolation_filter"]
    assert _initialiser(_text("oracle", "orc_mcstf.c"), "k_filter[16][8]") == [v for row in taps for v in row]
    assert all(row[0] == 0 and row[7] == 0 for row in taps)
    dev = _initialiser(_text("x265_amd", "csrc", "mcstf_kernels.hip"), "kMcstfFilter[16][6]")
    assert dev == [v for row in taps for v in row[1:7]]


def test_luma_filter_taps():
    taps = [v for row in GOLDEN["luma_filter"] for v in row]
    assert all(sum(row) == 64 for row in GOLDEN["luma_filter"])
    assert _initialiser(_text("oracle", "orc_ipfilter.c"), "k_luma_taps[4][8]") == taps
    assert _initialiser(_text("x265_amd", "csrc", "pu_kernels.hip"), "k_lumaTaps[4][8]") == taps
