"""Frame-segment sharding of the lookahead across ranks (one process per GPU).

Requests of different GOP segments are independent (a search reads the planes
of two frames, a cost evaluation those of three), so each rank owns a
contiguous frame segment and runs its whole request list locally; the only
exchange is a gather of the per-request scalars (costEst, costEstAq, intraMbs)
that the host-side decision logic consumes.  torch.distributed carries it: the
"nccl" backend is RCCL over xGMI on the GPU box, "gloo" on CPU in the tests.
"""
import numpy as np


def segment_bounds(total_frames, world, rank):
    """[start, end) of rank's contiguous segment; sizes differ by at most one frame."""
    base, extra = divmod(total_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def halo_bounds(total_frames, world, rank, reach):
    """Frames a rank must hold to run every request whose frame b lies in its segment of ONE lookahead
    window: the segment plus `reach` = bframes + 1 frames either side (a request reads frames b - d0 and
    b + d1 with distances up to bframes + 1).  Downscaling the halo frames again from their source is
    cheaper than shipping four planes per frame between GPUs."""
    start, end = segment_bounds(total_frames, world, rank)
    return max(0, start - reach), min(total_frames, end + reach)


def shard_requests(requests, start, end, origin=0):
    """The (p0, p1, b) requests of a window that belong to the rank owning frames [start, end), in their
    original order (all requests of one frame b stay together, so the request-order dependence of the
    searches is kept), with frame indices relative to `origin` (the first frame the rank holds).
    Returns (positions in `requests`, shifted requests)."""
    pos = [k for k, (_, _, b) in enumerate(requests) if start <= b < end]
    return pos, [(requests[k][0] - origin, requests[k][1] - origin, requests[k][2] - origin) for k in pos]


def gather_costs(local, device=None):
    """all_gather of an int64 [n, 3] array; returns one array per rank, in rank order.

    Segments may hold different request counts, so lengths are exchanged first and the
    payload is padded to the longest.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    local = np.ascontiguousarray(local, np.int64).reshape(-1, 3)
    dev = device if device is not None else "cpu"
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    buf = torch.zeros((m, 3), dtype=torch.int64, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [o[:c].cpu().numpy() for o, c in zip(out, counts)]
