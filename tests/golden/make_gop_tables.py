#!/usr/bin/env python3
"""Writes tests/golden/gop_ra.json: the data of x265's hierarchical mini-GOP tables (x265_gop_ra_length, x265_gop_ra and
x265_temporal_layer_bframes, source/x265.h:833-), read out of the reference's header where it lies.  A fixture of data
the reference holds, for tests/test_oracle_temporal_layers.py; run here, where /root/reference exists:

    python tests/golden/make_gop_tables.py
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = "/root/reference/source/x265.h"


def main():
    text = open(HEADER).read()
    at = text.index("x265_gop_ra[X265_MAX_GOP_CONFIG]")
    body = text[at:text.index("};", at)].split("=", 1)[1]
    nums = [int(v) for v in re.findall(r"-?\d+", body)]
    configs = int(re.search(r"#define X265_MAX_GOP_CONFIG (\d+)", text).group(1))
    length = int(re.search(r"#define X265_MAX_GOP_LENGTH (\d+)", text).group(1))
    assert len(nums) == configs * length * 3
    lengths = [int(v) for v in re.search(r"x265_gop_ra_length\[X265_MAX_GOP_CONFIG\] = \{([^}]*)\}", text).group(1).split(",")]
    bframes = [int(v) for v in re.search(r"x265_temporal_layer_bframes\[MAX_T_LAYERS\] = \{([^}]*)\}", text).group(1).split(",")]
    out = {"source": "source/x265.h", "gop_ra_length": lengths, "temporal_layer_bframes": bframes, "gop_ra": []}
    for g in range(configs):
        rows = [nums[(g * length + k) * 3:(g * length + k) * 3 + 3] for k in range(lengths[g])]
        out["gop_ra"].append({"poc_offset": [r[0] for r in rows], "layer": [r[1] for r in rows], "qp_offset": [r[2] for r in rows]})
    with open(os.path.join(HERE, "gop_ra.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
