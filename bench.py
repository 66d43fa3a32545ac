#!/usr/bin/env python3
"""Lookahead throughput on MI355X: frames/sec of x265's lowres lookahead, preset medium.

Two workloads per run, both through the C++ host layer (x265_amd/host) over the C-ABI:

  segment  (the headline `value`): one step = the whole lookahead work of one 60-frame segment handed to the
           device in x265's batch order (slicetype.cpp:2925-2988 on a fresh window): per frame Lowres::init,
           calcAdaptiveQuantFrame (AQ mode 2) and the intra estimate; batch 1 = every motion search (both lists,
           distances 1..bframes+1) with weightsAnalyse in front of each L0 search; batch 2 = every (p0, p1, b)
           frame cost; then one cuTree pass over the segment (B-pyramid mini-GOPs).  This is the throughput with
           many frames in flight, e.g. GOP segments of one or several encodes.  By default two lookahead
           instances per GPU (--in-flight 2: a library context, a stream and a host thread each) take the steps
           in turn, so that one segment's host phases (AQ mapping, weight analysis, cuTree) run beside the
           other's searches; every step is still a whole segment, and `single_instance` is the same step on one
           instance alone.
  steady   (`steady_state`): the real decision loop, Lookahead::slicetypeDecide with rc-lookahead 20 and a pool
           (batches per decision), fed picture by picture and drained as Encoder::encode does; frames/s and
           milliseconds per slicetypeDecide call are what ONE encoder instance sees.

The timed region is repeated (--reps, default 3): every repetition is exactly --steps steps between two fences;
`value` and `ms_per_step` are the median repetition's, `repetitions` lists them all.  `host_cpu` says what the host
spent beside the GPU (process CPU seconds per step, cores kept busy per GPU).  `other_configs` carries a few steps
each of the other sizes the metric is quoted on (4K 10-bit, 4K 10-bit --hme, 8K 10-bit).

Source pictures (Y, U, V) are resident in HBM before the timed region.  With --gpus N the parent starts N rank
processes (one per GPU, torch.distributed over RCCL); every rank runs its own segment (weak scaling) and the
per-request cost scalars of all ranks are gathered inside the timed step.  Under torchrun (WORLD_SIZE set) the
process is a rank itself.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
SIMDS = 1024            # 256 compute units x 4
ISSUE_BENCH = os.path.join(ROOT, "benchmarks", "micro", "issue_bench")
# Algorithmic integer operations of one 8x8 block compare, as lane operations: a SAD is 64 absolute differences
# accumulated, four (8-bit) or two (16-bit) per packed lane operation; a SATD is 64 subtractions and four 4x4
# Hadamards of 64 additions / subtractions, 16 absolute values and 15 additions each (~450); an 8x8 intra prediction
# is taken as 150 (64 samples of a two-tap interpolation with rounding, on average over DC / planar / angular).
OPS_SAD = {1: 16, 2: 32}
OPS_SATD = 450
OPS_INTRA_PRED = 150


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


def segment_slice_types(nframes, bframes):
    """The fixed mini-GOP pattern the segment workload's cuTree pass walks: I, then bframes B frames (the
    middle one a B-ref) before every P."""
    I, P, BREF, B = 2, 3, 4, 5
    types = [I]
    for k in range(1, nframes):
        pos = (k - 1) % (bframes + 1)
        types.append(P if pos == bframes else (BREF if bframes > 1 and pos == (bframes + 1) // 2 - 1 else B))
    types[-1] = P
    return types


def make_clip(width, height, frames, depth, seed):
    """(luma clip, [(u, v), ...]): the synthetic segment; chroma is only read by AQ."""
    from x265_amd import synth
    clip = synth.luma_clip(width, height, frames, depth, seed=seed)
    rng = np.random.default_rng(seed + 99)
    top = (1 << depth) - 1
    base = rng.integers(top // 3, 2 * top // 3, (height // 2, width // 2))
    chroma = []
    for i in range(frames):
        u = np.clip(np.roll(base, 2 * i, axis=1) + rng.integers(-3, 4, base.shape), 0, top).astype(clip.dtype)
        v = np.clip(np.roll(base[::-1], i, axis=0) + rng.integers(-3, 4, base.shape), 0, top).astype(clip.dtype)
        chroma.append((u, v))
    return clip, chroma


MEDIUM = dict(bframes=4, lookaheadDepth=20, bFrameAdaptive=2, bBPyramid=1, scenecutThreshold=40, scenecutBias=0.05,
              keyframeMin=25, keyframeMax=250, bOpenGOP=1, bIntraRefresh=0, radl=0, gopLookahead=0, cuTree=1, bFrameBias=0,
              vbvBufferSize=0, vbvMaxBitrate=0, bCqp=0, bEnableTemporalSubLayers=0, maxNumReferences=3,
              bEnableWeightedPred=1, bEnableWeightedBiPred=0, fpsNum=30, fpsDenom=1, qCompress=0.6, bitDepth=8,
              poolWorkers=16, bAdaptiveQuant=1, aqMode=2, aqStrength=1.0, bPresearch=1, lookaheadSlices=0)


# ---------------------------------------------------------------- CPU baseline: the oracle on host cores

def _oracle_frames(width, height, depth, bframes, clip, chroma, hme):
    from oracle import orc
    param = orc.make_param(width, height, bframes=bframes, weightp=True, hme=hme)
    look = orc.Lookahead(depth, param, with_aq=True)
    for i, y in enumerate(clip):
        qa, qc, inv, wsum, wssd = orc.calc_adaptive_quant_frame(depth, y, chroma[i][0], chroma[i][1], 2, 1.0, weightp=True)[:5]
        lr = look.add_frame(y, inv)
        lr.qp_aq_offset()[:] = qa
        lr.qp_cutree_offset()[:] = qc
        s, d = lr.wp()
        s[:], d[:] = wsum, wssd
    return look


def cpu_segment_seconds(width, height, depth, bframes, frames, seed, hme=False, counters=None):
    """The segment workload on the oracle, one core: seconds for `frames` frames.  `counters` (a dict) receives the
    block compares the oracle executed, per phase: the exact operation count of this clip's segment."""
    from oracle import orc
    clip, chroma = make_clip(width, height, frames, depth, seed)
    orc.op_counters(depth, reset=True)
    t0 = time.perf_counter()
    look = _oracle_frames(width, height, depth, bframes, clip, chroma, hme)
    b1, b2 = batch_requests(frames, bframes)
    for r in b1 + b2:
        look.frame_cost(*r)
    for k, t in enumerate(segment_slice_types(frames, bframes)):
        look.frames[k].set_slice_type(t)
    look.cu_tree(orc.CuTreeParam(30, 1, 1, 0, 20, 0, 0.6), frames - 1, False)
    dt = time.perf_counter() - t0
    look.close()
    if counters is not None:
        counters.update(orc.op_counters(depth))
    return dt


def _cpu_segment_worker(a):
    return cpu_segment_seconds(*a)


def cpu_steady_seconds(width, height, depth, frames, seed, hme=False):
    """The steady workload (slicetypeDecide loop, rc-lookahead 20, batches) on the oracle, one core."""
    from oracle import orc
    p = dict(MEDIUM, bitDepth=depth)
    clip, chroma = make_clip(width, height, frames, depth, seed)
    t0 = time.perf_counter()
    look = _oracle_frames(width, height, depth, p["bframes"], clip, chroma, hme)
    dec = orc.SliceDecider(look, p)
    for i in range(frames):
        dec.add_picture(i)
        dec.get_decided()
    dec.flush()
    while dec.get_decided() is not None:
        pass
    dt = time.perf_counter() - t0
    dec.close()
    look.close()
    return dt


def cpu_baselines(args):
    W, H, D = args.width, args.height, args.depth
    n = args.cpu_frames
    out = {}
    counters = {}
    dt = cpu_segment_seconds(W, H, D, args.bframes, n, 1234, args.hme, counters)
    out["_op_counters"] = {"frames": n, "phases": counters}
    out["cpu_baseline"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                           "sample": "segment workload, %d frames of %dx%d %d-bit, %.1f s on one core" % (n, W, H, D, dt)}
    dt = cpu_steady_seconds(W, H, D, n, 1234, args.hme)
    out["cpu_baseline_steady"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                                  "sample": "steady workload (slicetypeDecide loop, rc-lookahead 20), %d frames, %.1f s" % (n, dt)}
    cores = max(1, min(args.cpu_cores, os.cpu_count() or 1))
    if cores > 1:
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            pool.map(_cpu_segment_worker, [(W, H, D, args.bframes, n, 2000 + k, args.hme) for k in range(cores)])
        dt = time.perf_counter() - t0
        out["cpu_baseline_all_cores"] = {"value": cores * n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
                                         "sample": "segment workload, one independent %d-frame segment per core, %d cores, %.1f s"
                                                   % (n, cores, dt)}
    return out


# ---------------------------------------------------------------- measured machine constants and counters

def issue_rates():
    """Cycles per wave64 instruction and the clock held, measured on this GPU by benchmarks/micro/issue_bench."""
    if not os.path.exists(ISSUE_BENCH):
        return None
    try:
        txt = subprocess.run([ISSUE_BENCH], capture_output=True, timeout=120, check=True).stdout.decode()
        d = json.loads(txt)
    except (OSError, ValueError, subprocess.SubprocessError):
        return None
    return rates_from(d, "benchmarks/micro/issue_bench, run by this process (s_memtime / s_memrealtime)")


def recorded_issue_rates():
    """The microbenchmark's output as recorded on an MI355X (profiles/issue_bench_mi355x.json): used when this run may
    not start a child process (--no-micro, e.g. under rocprofv3)."""
    try:
        with open(os.path.join(ROOT, "profiles", "issue_bench_mi355x.json")) as f:
            return rates_from(json.load(f), "profiles/issue_bench_mi355x.json (recorded run of benchmarks/micro/issue_bench)")
    except (OSError, ValueError, KeyError):
        return None


def rates_from(d, source):
    k = d["kinds"]
    valu = [k[n]["waves_per_simd_8"]["cycles_per_instruction_per_simd"] for n in ("v_sad_u8", "v_pk_add_i16", "v_min_i32", "v_alignbyte_b32")]
    fastest = min(v["waves_per_simd_8"]["cycles_per_instruction_per_simd"] for n, v in k.items() if n.startswith("v_"))
    return {"valu_cycles_per_inst_per_simd_8_waves": sum(valu) / len(valu),
            "valu_cycles_per_inst_per_simd_8_waves_fastest_kind": fastest,
            "valu_cycles_per_inst_per_simd_1_wave": k["v_sad_u8"]["waves_per_simd_1"]["cycles_per_instruction_per_simd"],
            "salu_cycles_per_inst_per_simd_8_waves": k["s_add_u32"]["waves_per_simd_8"]["cycles_per_instruction_per_simd"],
            "clock_GHz": d["clock_GHz_at_8_waves"], "source": source}


def config_key(args):
    return "%dx%d_%dbit%s" % (args.width, args.height, args.depth, "_hme" if args.hme else "")


def pmc_kernel(args, kernel_name):
    """Counters of one kernel from the committed rocprofv3 --pmc passes of THIS configuration
    (profiles/pmc_<config>.json, made by profiles/run_pmc.sh + summarize_pmc.py), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_%s.json" % config_key(args))
    try:
        with open(path) as f:
            summary = json.load(f)
    except (OSError, ValueError):
        return None
    found = [c for name, c in summary.items() if kernel_name in name and c.get("dispatches")]
    if not found:
        return None
    if len(found) == 1:
        return found[0]
    # several instantiations of one kernel in a step (--hme: the quarter-res and the half-res search, one launch
    # each): per-step totals, every counter divided by its kernel's dispatch count first
    total = {"dispatches": 1.0}
    for c in found:
        for k, v in c.items():
            if k != "dispatches":
                total[k] = total.get(k, 0.0) + v / c["dispatches"]
    return total


def pmc_traffic_per_launch(c):
    """Bytes per launch between L2 and memory: (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch.  FETCH_SIZE counts
    128-byte requests as 64 bytes on gfx950, hence the factor 2 on the read side."""
    if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or not c.get("dispatches"):
        return None
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / c["dispatches"]


# ---------------------------------------------------------------- launcher

def launch_ranks(args, argv, script=None, timeout=None):
    """--gpus N outside torchrun: N fresh rank processes; nothing in this process touches the GPU.  Rank 0's
    standard output is passed on.  A rank that fails, or the whole set not finishing within `timeout` seconds,
    ends every remaining rank and the exit status is non-zero: a stuck rank must not hold the caller to its own
    limit."""
    script = script or os.path.abspath(__file__)
    timeout = args.rank_timeout if timeout is None else timeout
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = next((c for c in codes if c), 0)
            break
        failed = next((c for c in codes if c), 0)
        if failed or time.monotonic() > deadline:
            rc = failed or 124
            sys.stderr.write("bench.py: %s; ending the remaining ranks\n"
                             % ("a rank exited with status %d" % failed if failed else "ranks still running after %d s" % timeout))
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------- one configuration on the GPU

class Workload:
    """The segment workload of one configuration (size, depth, --hme) on this rank's GPU: source pictures resident in
    HBM, `in_flight` lookahead instances (library context, host-layer window, stream and host thread each)."""

    def __init__(self, cfg, rank, world, local_rank, distributed):
        import ctypes as C
        import torch
        from x265_amd import api
        self.C, self.torch, self.api = C, torch, api
        self.cfg, self.rank, self.world, self.local_rank, self.distributed = cfg, rank, world, local_rank, distributed
        W, H, D, F = cfg.width, cfg.height, cfg.depth, cfg.frames
        self.bpp = 1 if D == 8 else 2
        clip, chroma = cfg.clip if getattr(cfg, "clip", None) is not None else make_clip(W, H, F, D, 1234 + rank)

        def to_dev(a):
            return torch.from_numpy(a if D == 8 else a.view(np.int16)).cuda()
        self.dev_y = to_dev(clip)
        self.dev_u = to_dev(np.stack([c[0] for c in chroma]))
        self.dev_v = to_dev(np.stack([c[1] for c in chroma]))
        torch.cuda.synchronize()
        yb, cb = W * H * self.bpp, (W // 2) * (H // 2) * self.bpp
        self.y_ptrs = [self.dev_y.data_ptr() + i * yb for i in range(F)]
        self.u_ptrs = [self.dev_u.data_ptr() + i * cb for i in range(F)]
        self.v_ptrs = [self.dev_v.data_ptr() + i * cb for i in range(F)]

        hme_kw = dict(hme=True) if cfg.hme else {}
        table, centre = api.bitcost_table(D)
        self.Hlib = Hlib = api.load_host_library()
        self.b1, self.b2 = b1, b2 = batch_requests(F, cfg.bframes)
        a1 = np.ascontiguousarray(np.array(b1, np.int32).reshape(-1, 3))
        a2 = np.ascontiguousarray(np.array(b2, np.int32).reshape(-1, 3))
        self.n_search = sum(1 + (1 if p1 > b else 0) for (_, p1, b) in b1)
        self.n_cost = len(b1) + len(b2)
        vpn = C.c_void_p * F
        frames_arr = (C.c_int32 * F)(*range(F))
        ys, us, vs = vpn(*self.y_ptrs), vpn(*self.u_ptrs), vpn(*self.v_ptrs)
        sy, sc = (C.c_ssize_t * F)(*[W] * F), (C.c_ssize_t * F)(*[W // 2] * F)
        aqp = api.AqParams(2, 1, 0, 0, 1.0, 32, 0, None, None, None, 0, 0, 0.0, None)
        keep = self._keep = (a1, a2, frames_arr, ys, us, vs, sy, sc, aqp)

        class Pipeline:
            """One lookahead instance working on segments one after the other.  With --in-flight P, P of them take the
            steps in turn from P host threads, each on a stream of its own: one segment's host phases (weight analysis,
            AQ mapping, cuTree) then run beside another's searches."""

            def __init__(self, own_stream):
                self.stream = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
                self.ctx = api.Context(W, H, depth=D, bframes=cfg.bframes, num_slots=F, device=local_rank,
                                       stream=self.stream.cuda_stream, **hme_kw)
                self.ctx.set_mvcost(table, centre)
                n_ = self.ctx.geom.cuCount
                self.win = api.HostWindow(self.ctx, F, weightp=True)
                self.offs = [(np.zeros(n_), np.zeros(n_)) for _ in range(F)]
                self.invs = [np.zeros(n_, np.int32) for _ in range(F)]
                for i in range(F):
                    self.win.set_qp_offsets(i, *self.offs[i])
                for k, t in enumerate(segment_slice_types(F, cfg.bframes)):
                    self.win.set_slice_type(k, t)
                self.iq = vpn(*[a.ctypes.data for a in self.invs])

            def step(self):
                win = self.win
                win._chk(Hlib.xgh_pre_lookahead_aq(win.h, F, frames_arr, ys, us, vs, sy, sc, C.byref(aqp), self.iq, api.MEM_DEVICE))
                Hlib.xgh_add_many(win.h, len(a1), a1.ctypes.data_as(C.c_void_p))
                win.finish_batch()
                Hlib.xgh_add_many(win.h, len(a2), a2.ctypes.data_as(C.c_void_p))
                win.finish_batch()
                win.cu_tree(F - 1, False, b_pyramid=True)

            def close(self):
                self.win.close()
                self.ctx.close()

        self.in_flight = max(1, cfg.in_flight)
        self.pipes = [Pipeline(own_stream=self.in_flight > 1) for _ in range(self.in_flight)]
        self.ctx = self.pipes[0].ctx
        self.n = self.ctx.geom.cuCount
        self.req_all = np.ascontiguousarray(np.concatenate([a1, a2]))
        self.gather_lock = threading.Lock()
        if distributed:
            # per pipeline: the step's records in page-locked memory, their device copy, the gathered set
            for pipe in self.pipes:
                pipe.pinned = torch.zeros((self.n_cost, 3), dtype=torch.int64).pin_memory()
                pipe.pin_np = pipe.pinned.numpy()
                pipe.local = torch.zeros((self.n_cost, 3), dtype=torch.int64, device="cuda")
                pipe.gathered = torch.zeros((world * self.n_cost, 3), dtype=torch.int64, device="cuda")
                pipe.gathered_host = torch.zeros((world * self.n_cost, 3), dtype=torch.int64).pin_memory()

    def gather(self, pipe):
        # this rank's per-request scalars (already on its host, where its own decisions read them) go to the
        # other ranks in one collective; rank 0 also brings the whole set back to its host
        import torch.distributed as dist
        C = self.C
        self.Hlib.xgh_collect_costs(pipe.win.h, len(self.req_all), self.req_all.ctypes.data_as(C.c_void_p),
                                    pipe.pin_np.ctypes.data_as(C.c_void_p))
        pipe.local.copy_(pipe.pinned, non_blocking=True)
        dist.all_gather_into_tensor(pipe.gathered, pipe.local)
        if self.rank == 0:
            pipe.gathered_host.copy_(pipe.gathered, non_blocking=True)

    def segment_step(self, pipe=None):
        pipe = pipe or self.pipes[0]
        pipe.step()
        if self.distributed:
            # one collective at a time, in the same count on every rank
            with self.gather_lock:
                self.gather(pipe)

    def run_steps(self, count):
        """`count` segment steps: in turn on the pipelines, each pipeline driven by a host thread of its own."""
        torch = self.torch
        if self.in_flight == 1:
            for _ in range(count):
                self.segment_step()
            return
        errs = []

        def work(k):
            try:
                torch.cuda.set_device(self.local_rank)          # the current device is a per-thread setting
                with torch.cuda.stream(self.pipes[k].stream):
                    for _ in range(k, count, self.in_flight):
                        self.segment_step(self.pipes[k])
            except Exception as e:      # noqa: BLE001 - reported by the caller
                errs.append(e)
        th = [threading.Thread(target=work, args=(k,)) for k in range(self.in_flight)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    def fence(self):
        if self.distributed:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps):
        """Exactly `steps` steps between two fences -> (seconds: the slowest rank's, host CPU seconds of this process)."""
        self.fence()
        self.memo0 = self.ctx.aq_memo_stats()
        c0 = time.process_time()
        t0 = time.perf_counter()
        self.run_steps(steps)
        self.fence()
        dt = time.perf_counter() - t0
        cpu = time.process_time() - c0
        self.memo1 = self.ctx.aq_memo_stats()
        if self.distributed:
            import torch.distributed as dist
            t = self.torch.tensor([dt], dtype=self.torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, cpu

    def single_instance(self, steps):
        """One lookahead instance on its own (a segment's phases strictly one after the other), for comparison."""
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.segment_step()
        self.fence()
        sdt = time.perf_counter() - t0
        return {"ms_per_step": 1000.0 * sdt / steps, "frames_per_s_per_gpu": self.cfg.frames * steps / sdt, "steps": steps}

    def profile(self, prof_steps=2):
        """Per-phase kernel time with HIP events on the library's stream: a separate, untimed pass of one instance."""
        ctx = self.ctx
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        tp = time.perf_counter()
        for _ in range(prof_steps):
            self.segment_step()
        self.torch.cuda.synchronize()
        wall_ms = 1000.0 * (time.perf_counter() - tp) / prof_steps
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        return {k: (ms / prof_steps, launches / prof_steps) for k, (ms, launches) in prof.items()}, wall_ms

    def steady(self, steady_steps):
        """The real decision loop, one encoder instance: Lookahead::slicetypeDecide fed picture by picture."""
        api, cfg = self.api, self.cfg
        F, W = cfg.frames, cfg.width
        p = api.LookaheadParam(**dict(MEDIUM, bitDepth=cfg.depth, bframes=cfg.bframes))

        def steady_step():
            la = api.HostLookahead(self.ctx, p)
            out = 0
            for i in range(F):
                la.add_device_picture(self.y_ptrs[i], self.u_ptrs[i], self.v_ptrs[i], W, W // 2)
                out += la.get_decided() is not None
            la.flush()
            while la.get_decided() is not None:
                out += 1
            calls = la.decide_calls()
            for k, v in la.seconds().items():
                phases[k] = phases.get(k, 0.0) + v
            la.close()
            assert out == F
            return calls
        phases = {}
        steady_step()
        phases.clear()
        self.torch.cuda.synchronize()
        times, calls = [], 0
        for _ in range(steady_steps):
            t0 = time.perf_counter()
            calls += steady_step()
            self.torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        sdt = sum(times)
        med = sorted(times)[len(times) // 2]
        return {"frames_per_s": F / med, "ms_per_slicetypeDecide": 1000.0 * med / (calls / steady_steps),
                "slicetypeDecide_calls_per_segment": calls / steady_steps, "ms_per_segment": 1000.0 * med,
                "ms_per_segment_all": [1000.0 * t for t in times], "frames_per_s_mean": F * steady_steps / sdt,
                "ms_per_slicetypeDecide_by_phase": {k: 1000.0 * v / calls for k, v in phases.items()},
                "workload": "Lookahead::slicetypeDecide loop, preset medium (rc-lookahead 20, b-adapt 2, B-pyramid, weightp, AQ 2, "
                            "cuTree), pool of 16 (batched requests per decision), %d-frame clip fed picture by picture; the "
                            "median of %d passes" % (F, steady_steps)}

    def counted_cost_bytes(self):
        """Bytes the cost evaluation touches, counted from the vectors this step produced (frame k sits in slot k): a P
        request reads no pixels (vector costs, intra costs and AQ factors of each block); a B request reads the encode
        block, the two co-located blocks and, per list, the one half-pel plane block its vector points into, or the
        two blocks a quarter-pel vector averages (lowresMC, source/common/lowres.h:72-93)."""
        cache, n, ctx, g = {}, self.n, self.ctx, self.ctx.geom

        def planes(slot, lst, dist):
            if (slot, lst, dist) not in cache:
                mv = ctx.get_mvs(slot, lst, dist)[0]
                cache[(slot, lst, dist)] = int(n + np.count_nonzero((mv[:, 0] | mv[:, 1]) & 1))
            return cache[(slot, lst, dist)]
        total = 0
        for p0, p1, b in list(self.b1) + list(self.b2):
            total += 4 * g.heightInCU + 24                     # row sums and the request's totals
            if p1 > b:
                blocks = 3 * n + planes(b, 0, b - p0) + planes(b, 1, p1 - b)
                total += 64 * self.bpp * blocks + n * (8 + 8 + 4 + 4 + 4 + 2)
            else:
                total += n * (4 + 4 + 4 + 2)
        return total

    def close(self):
        for pipe in self.pipes:
            pipe.close()
        self.pipes = []
        self.dev_y = self.dev_u = self.dev_v = None
        self.torch.cuda.empty_cache()


def phase_report(wl, cfg, prof, rates, ops):
    """(kernels, roofline): per-phase kernel times with their algorithmic rates, and the dominant kernel's roofline.
    `ops`: executed block compares per phase from the oracle's run of the same segment, or None."""
    W, H, D, F = cfg.width, cfg.height, cfg.depth, cfg.frames
    bpp, n, g = wl.bpp, wl.n, wl.ctx.geom
    P = W * H
    n4 = g.widthInCU4 * g.heightInCU4 if cfg.hme else 0
    alg = {   # algorithmic bytes per step, per phase
        "lowres": F * 2 * P * bpp,
        "intra": F * ((P // 4) * bpp + 7 * n),
        "search": wl.n_search * (5 * (P // 4) * bpp + 12 * n),
        "cost": wl.counted_cost_bytes(),
    }
    kernels = {}
    for k, (ms_step, l_step) in prof.items():
        if ms_step <= 0:
            continue
        kernels[k] = {"ms_per_step": ms_step, "launches_per_step": l_step,
                      "avg_launch_us": 1000.0 * ms_step / max(1.0, l_step),
                      "algorithmic_GBs": alg[k] / (ms_step * 1e-3) / 1e9}
    dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    dom = kernels[dominant]
    mode = os.environ.get("X265GPU_SEARCH_MODE", "flow")
    search_kernel = {"flow": "search_flow_kernel", "rows": "search_rows_kernel"}.get(mode, "search_diag_kernel")
    names = {"lowres": "lowres_init_kernel", "intra": "intra_kernel", "search": search_kernel, "cost": "cost_rows_kernel"}
    kernel_name = names[dominant]
    # Launches of the dominant kernel in flight at a time: the diagonal mode spreads its job groups over concurrent
    # streams (X265GPU_SEARCH_STREAMS, default 2), so one launch lasts (streams x phase time / launches) while the
    # phase's aggregate rate is bytes / phase time; rocprofv3's per-kernel average duration is the former.  The flow
    # mode is one launch of the search kernel behind a small fill kernel, which the launch count leaves out.
    streams = 1
    dom_launches = dom["launches_per_step"]
    if dominant == "search" and mode == "diag":
        streams = max(1, min(4, int(os.environ.get("X265GPU_SEARCH_STREAMS", "2"))))
    elif dominant == "search" and mode == "flow":
        dom_launches = max(1.0, dom_launches / 2)
    bytes_per_launch = alg[dominant] / max(1.0, dom_launches)
    avg_launch_us = streams * 1000.0 * dom["ms_per_step"] / max(1.0, dom_launches)

    # Vector-issue fraction of every phase whose kernel has a counter pass for this configuration: wave64 VALU
    # instructions per step (counted per dispatch by rocprofv3, times the launches of a step) over the phase time,
    # against 1024 SIMDs x measured clock / measured cycles per instruction.  Two measured rates: the fastest kind of
    # vector instruction (plain 32-bit adds) is the ceiling no kernel can pass and the one fractions are quoted against;
    # the SAD / packed / DPP kind these kernels are mostly made of issues more slowly, and the fraction against that
    # rate says how much of the issue slots a kernel of this mix could still use.
    valu_peak = SIMDS * rates["clock_GHz"] / rates["valu_cycles_per_inst_per_simd_8_waves_fastest_kind"] if rates else None
    valu_peak_packed = SIMDS * rates["clock_GHz"] / rates["valu_cycles_per_inst_per_simd_8_waves"] if rates else None
    search_waves = wl.n_search * (flow_waves_per_job(g.widthInCU, g.heightInCU) +
                                  (flow_waves_per_job(g.widthInCU4, g.heightInCU4) if n4 else 0))
    for k, v in kernels.items():
        c = pmc_kernel(cfg, names[k])
        if not c or not valu_peak or not c.get("dispatches") or "SQ_INSTS_VALU" not in c:
            continue
        # launches of the phase's main kernel: the intra and cost phases count their small finalize kernel as well,
        # the flow search its fill kernel
        launches = v["launches_per_step"]
        if k in ("intra", "cost") or (k == "search" and mode == "flow"):
            launches /= 2
        if k == "search" and mode == "flow" and cfg.hme:
            launches = 1        # the two levels' kernels come back from pmc_kernel() summed per step
        insts = c["SQ_INSTS_VALU"] / c["dispatches"] * launches
        if k == "search" and mode == "flow" and c.get("SQ_WAVES"):
            # per wave rather than per dispatch: the counter pass may have run another segment length
            insts = c["SQ_INSTS_VALU"] / c["SQ_WAVES"] * search_waves
        v["valu_G_insts_per_s"] = insts / (v["ms_per_step"] * 1e-3) / 1e9
        v["valu_frac"] = v["valu_G_insts_per_s"] / valu_peak
        v["valu_frac_at_sad_class_rate"] = v["valu_G_insts_per_s"] / valu_peak_packed
        t = pmc_traffic_per_launch(c)
        if t is not None:
            v["pmc_traffic_bytes_per_step"] = t * launches
            v["pmc_traffic_over_algorithmic"] = t * launches / alg[k]

    # The algorithmic form of the same fraction: the block compares the reference's algorithm executes on this very
    # segment (counted by the oracle's run of it, cpu_baseline leg), as lane operations (OPS_* above), over 64 lanes x
    # the instruction peak x the phase time.  It does not grow when a kernel issues more instructions for the same work.
    if ops and valu_peak:
        lane_ops = {
            "intra": ops["intra"][1] * OPS_SATD + ops["intra"][2] * OPS_INTRA_PRED,
            "search": ops["search"][0] * OPS_SAD[bpp] + ops["search"][1] * OPS_SATD,
            "cost": ops["cost"][1] * OPS_SATD,
        }
        for k, total in lane_ops.items():
            if k in kernels and total:
                v = kernels[k]
                v["algorithmic_lane_ops_per_step"] = total
                v["algorithmic_frac"] = total / (64.0 * valu_peak * 1e9 * v["ms_per_step"] * 1e-3)
                v["algorithmic_frac_at_sad_class_rate"] = total / (64.0 * valu_peak_packed * 1e9 * v["ms_per_step"] * 1e-3)
        if "search" in kernels:
            units = wl.n_search * (n + n4)
            kernels["search"]["sads_per_block_search"] = ops["search"][0] / units
            kernels["search"]["satds_per_block_search"] = ops["search"][1] / units

    pmc = pmc_kernel(cfg, kernel_name)
    hbm = {"achieved": dom["algorithmic_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["algorithmic_GBs"] / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": bytes_per_launch, "traffic": pmc_traffic_per_launch(pmc)}
    roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hbm["frac"], "traffic": hbm["traffic"], "avg_launch_us": avg_launch_us, "concurrent_launches": streams,
                "launches_per_step": dom_launches, "algorithmic_bytes_per_launch": bytes_per_launch}
    if dominant == "search" and pmc and pmc.get("SQ_WAVES") and "SQ_INSTS_VALU" in pmc and rates:
        # The block search is integer SAD / SATD work out of LDS: the resource it uses most is the SIMDs' vector issue,
        # not HBM.  Instructions per block search come from the counter passes of this configuration, cycles per
        # instruction and the clock from the microbenchmark.
        units = wl.n_search * (n + n4)           # block searches per step; the launch starts a few more waves than
        waves = search_waves                     # that: chunks of a diagonal that are not full (they leave at once)
        per_search = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"] * waves / units
        achieved = per_search * units / (dom["ms_per_step"] * 1e-3) / 1e9
        roofline.update({"bound": "valu", "achieved": achieved, "peak": valu_peak, "unit": "G wave64-instructions/s",
                         "frac": achieved / valu_peak, "frac_at_sad_class_rate": achieved / valu_peak_packed,
                         "peak_at_sad_class_rate": valu_peak_packed, "valu_insts_per_block_search": per_search,
                         "salu_insts_per_block_search": pmc.get("SQ_INSTS_SALU", 0) / pmc["SQ_WAVES"] * waves / units,
                         "block_searches_per_step": units, "idle_waves_per_step": max(0.0, waves - units), "hbm": hbm,
                         "note": "vector-issue roofline: measured instructions per 8x8 block search x block searches / phase time, "
                                 "against 1024 SIMDs x measured clock / measured cycles per wave64 instruction of the fastest kind "
                                 "(32-bit add); frac_at_sad_class_rate uses the measured rate of v_sad_u8 / v_pk_add_i16 / v_min_i32 / "
                                 "v_alignbyte_b32, which the search is mostly made of and which issue 1.8x more slowly, measured for an "
                                 "independent stream; the kernel's stream is dependent (SAD, DPP adds, DPP minima, lane read per step) and "
                                 "by an occupancy test (8 -> 6 waves per SIMD costs 12 %) the phase is mostly at its issue limit; "
                                 "algorithmic_frac counts the reference algorithm's executed SADs / SATDs instead of issued instructions"})
        if "algorithmic_frac" in dom:
            roofline["algorithmic_frac"] = dom["algorithmic_frac"]
            roofline["algorithmic_frac_at_sad_class_rate"] = dom["algorithmic_frac_at_sad_class_rate"]
            roofline["algorithmic_lane_ops_per_launch"] = dom["algorithmic_lane_ops_per_step"] / max(1.0, dom_launches)
    elif dominant == "search":
        roofline["note"] = ("no counter pass committed for this configuration (profiles/pmc_%s.json): the vector-issue "
                            "fraction is not reported; the HBM fraction is small by construction (integer search out of LDS)"
                            % config_key(cfg))
    return kernels, roofline


def flow_waves_per_job(W8, H8):
    """Waves the single-launch search starts per job: four per chunk of an anti-diagonal t = x + 2y of the block grid
    (the last chunk of a diagonal may hold fewer than four blocks; its other waves leave at once)."""
    total = 0
    for t in range((W8 - 1) + 2 * (H8 - 1) + 1):
        ylo = max(0, (t - (W8 - 1) + 1) >> 1)
        yhi = min(H8 - 1, t >> 1)
        if yhi >= ylo:
            total += 4 * ((yhi - ylo + 1 + 3) // 4)
    return total


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


OTHER_CONFIGS = (   # the other sizes the metric is quoted on: (width, height, depth, hme, frames per step, steps)
    (3840, 2160, 10, False, 60, 3),
    (3840, 2160, 10, True, 60, 2),
    (7680, 4320, 10, False, 16, 3),
)


def other_configs(args, rates, rank, world, local_rank):
    """A bounded pass over the other configurations: a few steps each, frames/s, the search phase and its fractions."""
    out, clips = [], {}
    for (W, H, D, hme, F, steps) in OTHER_CONFIGS:
        cfg = argparse.Namespace(**dict(vars(args), width=W, height=H, depth=D, hme=hme, frames=F))
        key = (W, H, D, F)
        if key not in clips:
            clips.clear()                                   # one large clip in host memory at a time
            clips[key] = make_clip(W, H, F, D, 1234 + rank)
        cfg.clip = clips[key]
        wl = Workload(cfg, rank, world, local_rank, False)
        try:
            wl.run_steps(wl.in_flight)
            dt, _ = wl.timed(steps * wl.in_flight)
            prof, _ = wl.profile(1)
            kernels, roofline = phase_report(wl, cfg, prof, rates, None)
            rec = {"workload": "%dx%d %d-bit%s, %d-frame segment per step" % (W, H, D, " --hme" if hme else "", F),
                   "frames_per_s": F * steps * wl.in_flight / dt, "ms_per_step": 1000.0 * dt / (steps * wl.in_flight),
                   "steps": steps * wl.in_flight}
            for k in ("search", "cost", "intra", "lowres"):
                if k in kernels:
                    rec[k + "_ms"] = kernels[k]["ms_per_step"]
            if "search" in kernels:
                for k in ("valu_frac", "valu_frac_at_sad_class_rate", "pmc_traffic_over_algorithmic"):
                    if k in kernels["search"]:
                        rec["search_" + k] = kernels["search"][k]
            out.append(rec)
        finally:
            wl.close()
    return out


# ---------------------------------------------------------------- rank body

def rank_main(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Libraries under the process group print banners on stdout; keep fd 1 for the one JSON line.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # Host threads per lookahead instance for the library's double-precision AQ mapping: this rank's share of the host
    # (at most 16 cores per GPU, the share of a GPU on the boxes this runs on) split over its instances, so that N
    # ranks on one node do not start N x 2 x 16 mapping threads.
    in_flight = max(1, args.in_flight)
    share = max(1, min(16, len(os.sched_getaffinity(0)) // max(1, world)))
    host_threads = os.environ.get("X265GPU_HOST_THREADS")
    if host_threads is None:
        host_threads = str(max(1, share // in_flight))
        os.environ["X265GPU_HOST_THREADS"] = host_threads

    # Everything that starts another process (the issue-rate microbenchmark, the worker pool of the all-cores CPU
    # baseline) runs before this process initialises the GPU.
    rates = (issue_rates() if not args.no_micro else None) or recorded_issue_rates() if rank == 0 else None
    cpu = cpu_baselines(args) if rank == 0 and world == 1 and args.cpu_frames > 0 else {}
    ops = cpu.pop("_op_counters", None)
    if ops and ops["frames"] != args.frames:
        ops = None                      # the oracle's counts describe another segment length

    import torch
    distributed = world > 1 or args.force_dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the lookahead path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)       # --force-dist outside a launcher: a world of one
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W, H, D, F = args.width, args.height, args.depth, args.frames
    wl = Workload(args, rank, world, local_rank, distributed)
    wl.run_steps(max(args.warmup, in_flight if args.warmup else 0))
    reps = [wl.timed(args.steps) for _ in range(max(1, args.reps))]
    dts = [r[0] for r in reps]
    dt = median(dts)
    cpu_s = median([r[1] for r in reps])
    ms_per_step = 1000.0 * dt / args.steps
    fps = F * world * args.steps / dt
    gathered_ok = None
    if distributed and rank == 0:
        # the gathered set of the last step: this rank's own records sit at its place in it
        torch.cuda.synchronize()
        pipe = wl.pipes[(args.steps - 1) % in_flight] if in_flight > 1 else wl.pipes[0]
        mine = pipe.gathered_host.numpy()[rank * wl.n_cost:(rank + 1) * wl.n_cost]
        gathered_ok = bool(np.array_equal(mine, pipe.pin_np)) and bool(np.any(pipe.pin_np))

    memo_lookups, memo_hits = wl.memo1[0] - wl.memo0[0], wl.memo1[1] - wl.memo0[1]
    single = wl.single_instance(max(1, min(args.steps, 20))) if in_flight > 1 else None
    prof, prof_wall_ms = wl.profile(2)
    steady = wl.steady(args.steady_steps) if rank == 0 and args.steady_steps > 0 else None
    kernels, roofline = phase_report(wl, args, prof, rates, ops["phases"] if ops else None)
    n = wl.n
    n_search, n_cost = wl.n_search, wl.n_cost
    wl.close()

    others = None
    if rank == 0 and world == 1 and not args.force_dist and not args.no_other_configs:
        others = other_configs(args, rates, rank, world, local_rank)

    out = {
        "metric": "lookahead frames/sec", "value": fps, "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8" if D == 8 else "u16", "data": "synthetic",
        "config": {"workload": "%dx%d %d-bit lookahead%s, preset medium per segment: Lowres::init + AQ mode 2 + intra per frame, "
                               "batch 1 (all searches, weightp analysis) + batch 2 (all frame costs) in b-adapt 2 batch order, one cuTree "
                               "pass; %d-frame segment per step%s" % (W, H, D, " with --hme (hex / umh)" if args.hme else "", F,
                                                                   ", %d lookahead instances per GPU taking the steps in turn" % in_flight
                                                                   if in_flight > 1 else ""),
                   "frames_per_step": F, "search_requests_per_step": n_search, "cost_requests_per_step": n_cost,
                   "blocks_per_frame": n, "bframes": args.bframes,
                   "segments_in_flight_per_gpu": in_flight,
                   "sharding": "one frame segment per rank, RCCL all_gather of the per-request cost scalars" if distributed else "single GPU"},
        "repetitions": {"count": len(dts), "steps_each": args.steps, "ms_per_step": [1000.0 * t / args.steps for t in dts],
                        "frames_per_s": [F * world * args.steps / t for t in dts],
                        "ms_per_step_median": ms_per_step, "ms_per_step_min": 1000.0 * min(dts) / args.steps,
                        "ms_per_step_max": 1000.0 * max(dts) / args.steps,
                        "note": "value and ms_per_step are the median repetition; every repetition is exactly `steps` steps between two fences"},
        "host_cpu": {"host_cpu_seconds_per_step": cpu_s / args.steps, "host_cores_needed_per_gpu": cpu_s / dt,
                     "host_threads_per_instance": int(host_threads), "host_cores_visible": len(os.sched_getaffinity(0)),
                     "aq_activity_memo_hit_rate": memo_hits / memo_lookups if memo_lookups else None,
                     "note": "process CPU time of this rank over the timed region (every thread: the instances' host threads and the "
                             "library's AQ mapping threads) / wall time = cores this rank keeps busy beside its GPU"},
        "roofline": roofline, "kernels": kernels, "machine": rates,
        "step_wall_ms_with_event_timers": prof_wall_ms,
    }
    if gathered_ok is not None:
        out["gathered_records_match_local"] = gathered_ok
    if single:
        out["single_instance"] = single
    if steady:
        out["steady_state"] = steady
    if others is not None:
        out["other_configs"] = others
    out.update(cpu)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3, help="repetitions of the timed block of --steps steps; the median is reported")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--frames", type=int, default=60, help="frames per step (one segment)")
    ap.add_argument("--bframes", type=int, default=4)
    ap.add_argument("--in-flight", type=int, default=2,
                    help="segments in flight per GPU: lookahead instances (context, stream, host thread) that take the steps in turn")
    ap.add_argument("--steady-steps", type=int, default=3, help="repetitions of the steady (slicetypeDecide loop) workload, 0 = skip")
    ap.add_argument("--cpu-frames", type=int, default=60, help="frames in the bounded CPU baseline samples (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=16, help="host cores for the all-cores CPU baseline")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the gather even at world size 1 (rehearsal of the N>1 path)")
    ap.add_argument("--hme", action="store_true", help="--hme: quarter-res hex then half-res umh (x265's defaults, ranges 16 / 32)")
    ap.add_argument("--no-micro", action="store_true", help="skip the issue-rate microbenchmark (it runs as a child process)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the bounded pass over 4K / 4K --hme / 8K")
    ap.add_argument("--rank-timeout", type=int, default=900, help="--gpus N launcher: seconds after which remaining ranks are ended")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank_main(args)


if __name__ == "__main__":
    main()
