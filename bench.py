#!/usr/bin/env python3
"""Lookahead throughput on MI355X: frames/sec of the lowres lookahead hot path.

One step = the whole lookahead workload of one synthetic clip segment, issued
the way x265's batch mode issues it (slicetype.cpp:2925-2988): Lowres::init and
the intra estimate for every frame, then batch 1 (every motion search, both
lists, distances 1..bframes+1) and batch 2 (every (p0, p1, b) frame-cost
combination).  Source frames are resident in HBM before the timed region; the
per-request scalars come back to the host inside it.  With --gpus N every rank
runs its own segment (weak scaling) and the cost scalars of all ranks are
gathered over RCCL inside the timed step.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak


def batch_requests(nframes, bframes):
    """(batch1, batch2) as lists of (p0, p1, b): the two batch loops of slicetypeAnalyse
    (slicetype.cpp:2925-2988) on a fresh window, including their skip conditions (a search is
    queued once per (frame, list, distance); batch 2 only measures combinations whose searches
    were queued in batch 1)."""
    searched = [[set(), set()] for _ in range(nframes)]     # per frame: distances searched per list
    done = [set() for _ in range(nframes)]                  # per frame: (d0, d1) costs measured
    b1, b2 = [], []
    for b in range(2, nframes):
        for i in range(1, bframes + 2):
            p0 = b - i
            if p0 < 0 or i in searched[b][0]:
                continue
            p1 = b + i
            if p1 >= nframes or i in searched[b][1]:
                p1 = b
            b1.append((p0, p1, b))
            searched[b][0].add(i)
            if p1 > b:
                searched[b][1].add(i)
            done[b].add((i, p1 - b))
    for b in range(2, nframes):
        for i in range(1, bframes + 2):
            if b < i or i not in searched[b][0]:
                continue
            for j in range(0, bframes + 1):
                if b + j >= nframes:
                    break
                if j and j not in searched[b][1]:
                    continue
                if (i, j) in done[b]:
                    continue
                b2.append((b - i, b + j, b))
                done[b].add((i, j))
    return b1, b2


def plan_step(ctx, nframes, bframes):
    """Device requests of one step with their doSearch flags (identical for every fresh window)."""
    from x265_amd.lookahead import CostEstimateGroup, LowresState
    frames = [LowresState(i, i, ctx.D) for i in range(nframes)]
    est = CostEstimateGroup(ctx, frames)
    b1, b2 = batch_requests(nframes, bframes)
    r1 = [est._request(*r) for r in b1]
    for fr in frames:
        fr.pending.clear()
    r2 = [est._request(*r) for r in b2]
    assert None not in r1 and None not in r2
    assert not any(r[6] or r[7] for r in r2), "batch 2 must not search"
    return r1, r2


def cpu_baseline(width, height, depth, bframes, sample_frames, seed, hme=False):
    """The oracle (a scalar C port of the reference path) on one host core, same request pattern."""
    from oracle import orc
    from x265_amd import synth
    clip = synth.luma_clip(width, height, sample_frames, depth, seed=seed)
    param = orc.make_param(width, height, bframes=bframes, hme=hme)
    t0 = time.perf_counter()
    ref = orc.Lookahead(depth, param)
    for f in clip:
        ref.add_frame(f)
    b1, b2 = batch_requests(sample_frames, bframes)
    for r in b1 + b2:
        ref.frame_cost(*r)
    dt = time.perf_counter() - t0
    ref.close()
    return {"value": sample_frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d-frame %dx%d %d-bit window, same batch-1 + batch-2 request pattern, %.1f s"
                      % (sample_frames, width, height, depth, dt)}


PMC_SUMMARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "latest_pmc_summary.json")


def _pmc_kernel(kernel_name):
    """Counters of one kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/run_pmc.sh + summarize_pmc.py), or None."""
    try:
        with open(PMC_SUMMARY) as f:
            summary = json.load(f)
    except (OSError, ValueError):
        return None
    for name, counters in summary.items():
        if kernel_name in name:
            return counters
    return None


def pmc_traffic_per_launch(kernel_name):
    """Bytes per launch between L2 and memory: (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch.  FETCH_SIZE counts
    128-byte requests as 64 bytes on gfx950, hence the factor 2 on the read side."""
    c = _pmc_kernel(kernel_name)
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or not c.get("dispatches"):
        return None
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / c["dispatches"]


def valu_utilisation(kernel_name, units, ms_per_step):
    """Share of the chip's VALU issue slots the dominant kernel fills: wave-level VALU instructions per step
    (counter pass: SQ_INSTS_VALU per wave x waves) x 4 cycles each / (1024 SIMDs x clock x time)."""
    c = _pmc_kernel(kernel_name)
    if not c or not units or not c.get("SQ_WAVES") or "SQ_INSTS_VALU" not in c:
        return None
    per_wave = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    clock_hz = 2.4e9
    cycles = per_wave * units * 4.0 / 1024.0
    return {"valu_insts_per_block_search": per_wave, "salu_insts_per_block_search": c.get("SQ_INSTS_SALU", 0) / c["SQ_WAVES"],
            "frac_of_valu_issue_at_2.4GHz": cycles / (clock_hz * ms_per_step * 1e-3),
            "note": "one wave = one 8x8 block search; instruction counts from profiles/latest_pmc_summary.json"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--frames", type=int, default=60, help="frames per step (one device batch)")
    ap.add_argument("--bframes", type=int, default=4)
    ap.add_argument("--cpu-frames", type=int, default=60, help="frames in the bounded CPU baseline sample (0 = skip)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the gather even at world size 1 (rehearsal of the N>1 path)")
    ap.add_argument("--hme", action="store_true", help="--hme: quarter-res hex then half-res umh (x265's defaults, ranges 16 / 32)")
    args = ap.parse_args()

    # Libraries under the process group print banners on stdout; keep fd 1 for the one JSON line.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or args.force_dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the lookahead path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from oracle import orc
    from x265_amd import api, synth

    W, H, D, F = args.width, args.height, args.depth, args.frames
    bpp = 1 if D == 8 else 2
    clip = synth.luma_clip(W, H, F, D, seed=1234 + rank)
    if D == 8:
        dev_clip = torch.from_numpy(clip).cuda()
    else:
        dev_clip = torch.from_numpy(clip.view(np.int16)).cuda()
    torch.cuda.synchronize()
    frame_bytes = W * H * bpp
    ptrs = [dev_clip.data_ptr() + i * frame_bytes for i in range(F)]
    slots = list(range(F))

    stream = torch.cuda.current_stream().cuda_stream
    ctx = api.Context(W, H, depth=D, bframes=args.bframes, num_slots=F, device=local_rank, stream=stream, hme=args.hme)
    table, centre = orc.bitcost_table(D)          # host-built input table (float math stays on the host)
    ctx.set_mvcost(table, centre)
    r1, r2 = plan_step(ctx, F, args.bframes)
    a1, a2 = ctx._pack(r1), ctx._pack(r2)
    o1, o2 = (api.FrameCost * len(a1))(), (api.FrameCost * len(a2))()
    n_search = sum(int(r[6]) + int(r[7]) for r in r1 + r2)
    n_cost = len(r1) + len(r2)
    gather_buf = None
    if distributed:
        local = torch.zeros((n_cost, 3), dtype=torch.int64, device="cuda")
        cost_dtype = np.dtype([("costEst", "<i8"), ("costEstAq", "<i8"), ("intraMbs", "<i4"), ("reserved", "<i4")])
        host = np.zeros((n_cost, 3), np.int64)
        offs = [0, len(a1), len(a1) + len(a2)]
        gather_buf = [torch.zeros_like(local) for _ in range(world)]

    def step():
        ctx.lowres_init_device(slots, ptrs, [W] * F)
        ctx.intra_estimate(slots)
        ctx._chk(ctx.L.x265gpu_estimate_costs(ctx.h, len(a1), a1, o1))
        ctx._chk(ctx.L.x265gpu_estimate_costs(ctx.h, len(a2), a2, o2))
        if distributed:
            for k, o in enumerate((o1, o2)):        # x265gpu_frame_cost records viewed in place, no per-record Python work
                rec = np.frombuffer(o, dtype=cost_dtype)
                host[offs[k]:offs[k + 1], 0] = rec["costEst"]
                host[offs[k]:offs[k + 1], 1] = rec["costEstAq"]
                host[offs[k]:offs[k + 1], 2] = rec["intraMbs"]
            local.copy_(torch.from_numpy(host))
            dist.all_gather(gather_buf, local)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1000.0 * dt / args.steps
    fps = F * world * args.steps / dt

    # per-phase kernel time with HIP events on the library's stream (separate, untimed pass)
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    prof_steps = 2
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)

    g = ctx.geom
    P = W * H
    n = g.cuCount
    alg = {   # algorithmic bytes per step, per phase
        "lowres": F * 2 * P * bpp,
        "intra": F * ((P // 4) * bpp + 7 * n),
        "search": n_search * (5 * (P // 4) * bpp + 12 * n),
        "cost": n_cost * (9 * (P // 4) * bpp + 2 * n + 4 * g.heightInCU + 20),
    }
    dominant = max(prof, key=lambda k: prof[k][0])
    kernels = {}
    for k, (ms, launches) in prof.items():
        ms_step, l_step = ms / prof_steps, launches / prof_steps
        if ms_step <= 0:
            continue
        kernels[k] = {"ms_per_step": ms_step, "launches_per_step": l_step,
                      "avg_launch_us": 1000.0 * ms_step / max(1.0, l_step),
                      "achieved_GBs": alg[k] / (ms_step * 1e-3) / 1e9}
    dom = kernels[dominant]
    kernel_name = {"lowres": "lowres_init_kernel", "intra": "intra_kernel", "search": "search_diag_kernel",
                   "cost": "cost_rows_kernel"}[dominant]
    # The search spreads its job groups over concurrent streams (X265GPU_SEARCH_STREAMS, default 2): that many
    # launches are in flight at a time, so one launch lasts (streams x phase time / launches) while the phase's
    # aggregate rate is bytes / phase time.  rocprofv3's per-kernel average duration is the former.
    streams = 1
    if dominant == "search" and os.environ.get("X265GPU_SEARCH_MODE", "diag") != "rows":
        streams = max(1, min(4, int(os.environ.get("X265GPU_SEARCH_STREAMS", "2"))))
    bytes_per_launch = alg[dominant] / max(1.0, dom["launches_per_step"])
    avg_launch_us = streams * dom["avg_launch_us"]
    roofline = {"bound": "hbm", "kernel": kernel_name,
                "achieved": streams * bytes_per_launch / (avg_launch_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["achieved_GBs"] / HBM_PEAK_GBS, "traffic": pmc_traffic_per_launch(kernel_name),
                "avg_launch_us": avg_launch_us, "concurrent_launches": streams,
                "launches_per_step": dom["launches_per_step"], "algorithmic_bytes_per_launch": bytes_per_launch,
                "note": "the integer SAD/SATD search is VALU-issue bound, not HBM bound (see 'valu'); "
                        "per-phase figures in 'kernels'"}
    valu = valu_utilisation(kernel_name, n_search * n if dominant == "search" else None, dom["ms_per_step"])

    out = {
        "metric": "lookahead frames/sec", "value": fps, "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8" if D == 8 else "u16", "data": "synthetic",
        "config": {"workload": "%dx%d %d-bit lookahead%s, bframes %d, b-adapt 2 batch order, %d-frame segment per step"
                               % (W, H, D, " with --hme (hex / umh)" if args.hme else "", args.bframes, F),
                   "frames_per_step": F, "search_requests_per_step": n_search, "cost_requests_per_step": n_cost,
                   "blocks_per_frame": n, "sharding": "frame segments per rank, RCCL all_gather of cost scalars"
                   if distributed else "single GPU"},
        "roofline": roofline, "valu": valu, "kernels": kernels,
    }
    if rank == 0 and args.cpu_frames > 0:
        out["cpu_baseline"] = cpu_baseline(W, H, D, args.bframes, args.cpu_frames, 1234, args.hme)
    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
