#!/usr/bin/env python3
"""Writes tests/golden/reference_tables.json: the constant data the lookahead path reads in the reference, read out of
the reference's sources where they lie (numbers only, no code):

  search patterns hex2, mod6m1, square1, hex4            source/encoder/motion.cpp:64-73
  x265_lambda_tab at X265_LOOKAHEAD_QP, per bit depth    source/common/constants.cpp:30-130, common.h:217-221
  x265_exp2_lut                                          source/common/constants.cpp:552-558
  g_intraFilterFlags                                     source/common/constants.cpp:561-567
  g_lumaFilter                                           source/common/constants.cpp:250-256
  angleTable, invAngleTable                              source/common/intrapred.cpp:123-124
  s_interpolationFilter                                  source/common/temporalfilter.h:36-54

A fixture of data the reference holds, for tests/test_reference_tables.py; run here, where /root/reference exists:

    python tests/golden/make_reference_tables.py
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/source"
NUM = r"-?(?:0x[0-9a-fA-F]+|\d+\.\d+|\d+)"


def body(text, name):
    """The text between the braces of the initialiser that follows `name`."""
    at = text.index(name)
    a = text.index("=", at)
    b = text.index("};", a)
    return re.sub(r"//[^\n]*|/\*.*?\*/", "", text[a + 1:b], flags=re.S)


def ints(text, name):
    return [int(v, 0) for v in re.findall(NUM, body(text, name))]


def mvs(text, name):
    return [[int(x), int(y)] for x, y in re.findall(r"MV\((-?\d+),\s*(-?\d+)\)", body(text, name))]


def main():
    motion = open(os.path.join(REF, "encoder/motion.cpp")).read()
    consts = open(os.path.join(REF, "common/constants.cpp")).read()
    common = open(os.path.join(REF, "common/common.h")).read()
    intra = open(os.path.join(REF, "common/intrapred.cpp")).read()
    tf = open(os.path.join(REF, "common/temporalfilter.h")).read()

    out = {"source": "source/encoder/motion.cpp, source/common/{constants.cpp,common.h,intrapred.cpp,temporalfilter.h}"}
    out["hex2"] = mvs(motion, "const MV hex2[8]")
    out["mod6m1"] = ints(motion, "mod6m1[8]")
    out["square1"] = mvs(motion, "const MV square1[9]")
    out["hex4"] = mvs(motion, "const MV hex4[16]")

    # the three x265_lambda_tab bodies follow '#if X265_DEPTH == 12', '#elif X265_DEPTH == 10', '#else' in this order
    qp_base = int(re.search(r"#define X265_LOOKAHEAD_QP \((\d+) \+ QP_BD_OFFSET\)", common).group(1))
    per_bit = int(re.search(r"#define QP_BD_OFFSET \((\d+) \* \(X265_DEPTH - 8\)\)", common).group(1))
    tabs = [m.start() for m in re.finditer(r"double x265_lambda_tab\[", consts)]
    assert len(tabs) == 3
    out["lookahead_qp"] = {}
    out["lookahead_lambda"] = {}
    for depth, at in zip((12, 10, 8), tabs):
        vals = [float(v) for v in re.findall(NUM, body(consts[at:], "x265_lambda_tab[QP_MAX_MAX + 1]"))]
        qp = qp_base + per_bit * (depth - 8)
        out["lookahead_qp"][str(depth)] = qp
        out["lookahead_lambda"][str(depth)] = vals[qp]

    out["exp2_lut"] = ints(consts, "x265_exp2_lut[64]")
    out["intra_filter_flags"] = ints(consts, "g_intraFilterFlags[NUM_INTRA_MODE]")
    luma = ints(consts, "g_lumaFilter[4][NTAPS_LUMA]")
    assert len(luma) == 32
    out["luma_filter"] = [luma[8 * k:8 * k + 8] for k in range(4)]
    out["angle_table"] = ints(intra, "angleTable[17]")
    out["inv_angle_table"] = ints(intra, "invAngleTable[8]")
    flt = ints(tf, "s_interpolationFilter[16][8]")
    assert len(flt) == 128
    out["interpolation_filter"] = [flt[8 * k:8 * k + 8] for k in range(16)]

    assert len(out["hex2"]) == 8 and len(out["mod6m1"]) == 8 and len(out["square1"]) == 9 and len(out["hex4"]) == 16
    assert len(out["exp2_lut"]) == 64 and len(out["intra_filter_flags"]) == 35
    assert len(out["angle_table"]) == 17 and len(out["inv_angle_table"]) == 8
    with open(os.path.join(HERE, "reference_tables.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
