"""Throughput of the --threaded-me job generator (x265_amd/host/threaded_me_host: Analysis::deriveMVsForCTU for whole
CTU rows): every enabled PU shape of every CTU of a 3840x2160 picture, star search, merange 57, subme 3 (BASELINE
config 4), predictors from the CTU's own table, the lookahead's vectors as extra candidates.

Prints one JSON line.  `--cpu-ctus N` times the serial restatement over the CPU oracle (tests/tme_serial.py, test
infrastructure) on the first N CTUs of a row and checks them against the GPU table.

    python benchmarks/tme_bench.py [--b-picture] [--amp] [--rows 34] [--cpu-ctus 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="CTU rows per call (0: the whole picture)")
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--b-picture", action="store_true")
    ap.add_argument("--cpu-ctus", type=int, default=2)
    ap.add_argument("--in-flight", type=int, default=1,
                    help="generator instances (library context + ThreadedME + host thread each) that take the row groups of a "
                         "picture in turn: one's host stages run beside another's device calls (CTUs are independent)")
    a = ap.parse_args()
    W, H, ctu, lam = a.width, a.height, 64, 4.0
    n = 3 if a.b_picture else 2
    clip = synth.luma_clip(W, H, n, a.depth, seed=1234, scene_cut=False)
    cur = 1
    from oracle import orc          # host-built input tables, as BitCost makes them
    la_table, la_centre = orc.bitcost_table(a.depth)
    import threading
    P = max(1, a.in_flight)
    ctxs, tmes = [], []
    for _ in range(P):
        c = api.Context(W, H, depth=a.depth, num_slots=n, ctu=ctu, keep_full_res=True, bframes=4)
        for k in range(n):
            c.lowres_init(k, clip[k], poc=k)
        c.set_mvcost(la_table, la_centre)
        c.intra_estimate(list(range(n)))
        c.estimate([(0, cur, cur, 0, cur, cur, 1, 0)])
        if a.b_picture:
            c.estimate([(0, cur, 2, 0, cur, 2, 0, 1)])
        ctxs.append(c)
        tmes.append(api.ThreadedME(c, search=api.STAR_SEARCH, merange=57, subpel=3, amp=a.amp, rect=True, min_cu=8, lam=lam))
    ctx, tme = ctxs[0], tmes[0]
    lowres0 = ctx.get_mvs(cur, 0, 1)[0]
    refs = [(0, 0, True, None, lowres0)]
    refs_l1 = [(2, 2, True, None, ctx.get_mvs(cur, 1, 1)[0])] if a.b_picture else []
    counts = tme.counts()
    rows_total = counts["rows"]
    step = a.rows or -(-rows_total // P)
    table = tme.new_table()
    for t in tmes:
        t.estimate_rows(cur, cur, refs, 0, 1, table, refs_l1=refs_l1)          # warm-up
    groups = list(range(0, rows_total, step))

    def work(k):
        for r in groups[k::P]:
            tmes[k].estimate_rows(cur, cur, refs, r, min(rows_total, r + step), table, refs_l1=refs_l1)
    times = []
    for _ in range(a.reps):
        s0 = [t.counts() for t in tmes]
        sec0 = [t.seconds() for t in tmes]
        t0 = time.perf_counter()
        if P == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(k,)) for k in range(P)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        times.append(time.perf_counter() - t0)
        s1 = [t.counts() for t in tmes]
        sec1 = [t.seconds() for t in tmes]
    dt = sorted(times)[len(times) // 2]
    searches = sum(b["searches"] - a_["searches"] for a_, b in zip(s0, s1))
    calls = sum(b["device_calls"] - a_["device_calls"] for a_, b in zip(s0, s1))
    sec0 = {k: sum(x[k] for x in sec0) for k in sec0[0]}
    sec1 = {k: sum(x[k] for x in sec1) for k in sec1[0]}
    ctus = counts["cols"] * rows_total
    out = {"metric": "--threaded-me pictures per second (star, merange 57, subme 3)", "value": 1.0 / dt, "unit": "pictures/s",
           "seconds_per_picture": dt, "ctus": ctus, "pu_searches_per_picture": searches, "pu_searches_per_s": searches / dt,
           "device_calls_per_picture": calls, "host_logic_s": sec1["host"] - sec0["host"],
           "inside_device_calls_s": sec1["device"] - sec0["device"], "tasks_per_ctu": counts["tasks"], "dependency_levels": counts["levels"],
           "config": {"workload": "%dx%d %d-bit %s picture, %s PU shapes, min CU 8, %d CTU rows per call"
                                  % (W, H, a.depth, "B (one reference each way)" if a.b_picture else "P (one reference)",
                                     "all 24" if a.amp else "square and rectangular", step),
                      "generator_instances": P},
           "note": "wall time of ThreadedME::estimateRows over the whole picture: host predictor logic, job upload, kernels, "
                   "result download, level after level"}
    if a.cpu_ctus > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from tme_serial import SerialTME
        mx, my = ctu + 32, ctu + 16
        fw, fh = -(-W // ctu) * ctu, -(-H // ctu) * ctu
        pads = [np.pad(f, ((my, my + fh - H), (mx, mx + fw - W)), mode="edge") for f in clip]
        table_me, centre_me = orc.bitcost_table(a.depth, lam)
        frames = [0] + ([2] if a.b_picture else [])
        mes = [orc.FullResME(a.depth, table_me, centre_me, pads[cur], pads[f], mx, my) for f in frames]
        ser = SerialTME(mes, frames, cur, W, H, orc.bitsizes(a.depth), (table_me, centre_me), lam, method=api.STAR_SEARCH,
                        merange=57, subpel=3, amp=a.amp, rect=True, ctu=ctu, min_cu=8,
                        lowres_mvs=[lowres0] + [r[4] for r in refs_l1], num_l0=1)
        row = rows_total // 2
        t0 = time.perf_counter()
        for addr in range(row * counts["cols"], row * counts["cols"] + a.cpu_ctus):
            ser.derive_mvs_for_ctu(addr)
        cpu_dt = time.perf_counter() - t0
        lo = row * counts["cols"]
        for name in ("ref", "mv", "mvp", "mvCost", "bits", "cost"):
            np.testing.assert_array_equal(table[name][lo:lo + a.cpu_ctus], ser.table[name][lo:lo + a.cpu_ctus], err_msg=name)
        out["cpu_baseline"] = {"value": a.cpu_ctus / cpu_dt / ctus, "unit": "pictures/s", "cores": 1, "kind": "port",
                               "sample": "%d CTUs of row %d, serial restatement over the oracle's searches (the searches are C, "
                                         "the bookkeeping Python), checked equal to the GPU table" % (a.cpu_ctus, row)}
        for m in mes:
            m.close()
    for t in tmes:
        t.close()
    for c in ctxs:
        c.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
