"""GPU: x265gpu_weight_cost_static, the entry point that allocates nothing, against x265gpu_weight_cost.

Both run weightCostLuma (source/encoder/slicetype.cpp:817-850) on the same slots; the static form keeps its
accumulator and the weighted copy of plane 0 in the library's fixed device array, which every context on a GPU
shares and which keeps the previous call's contents.  Two calls in a row, the second on a smaller picture, must
both give the allocating entry point's result exactly (integer sums: bit-exact)."""
import numpy as np
import pytest

from x265_amd import api, synth

pytestmark = pytest.mark.gpu


def _prepare(ctx, W, H, depth, seed):
    clip = synth.luma_clip(W, H, 2, depth, seed=seed)
    for i, f in enumerate(clip):
        ctx.lowres_init(i, f, i)
    ctx.intra_estimate([0, 1])
    ctx.sync()


WEIGHTS = [(0, 0, 0, 0), (1, 61, 6, 3), (1, 127, 7, -9), (1, 40, 5, 20)]


@pytest.mark.parametrize("depth", [8, 10])
def test_static_scratch_matches_allocating_entry_point(depth):
    big = api.Context(1280, 720, depth=depth, num_slots=2)
    small = api.Context(416, 240, depth=depth, num_slots=2)
    try:
        _prepare(big, 1280, 720, depth, 5)
        _prepare(small, 416, 240, depth, 6)
        want_big = [big.weight_cost(1, 0, *w) for w in WEIGHTS]
        want_small = [small.weight_cost(1, 0, *w) for w in WEIGHTS]
        assert len(set(want_big)) > 1 and min(want_big) > 0
        # first call: the larger picture; second call: the smaller one, whose plane covers only the start of
        # what the first left in the shared array
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")        # the runtime the library under test already runs on

        def free_bytes():
            free, total = C.c_size_t(), C.c_size_t()
            assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
            return free.value
        free_before = free_bytes()
        for w, wb, ws in zip(WEIGHTS, want_big, want_small):
            assert big.weight_cost_static(1, 0, *w) == wb, w
            assert small.weight_cost_static(1, 0, *w) == ws, w
        # "allocates nothing": the device's free memory is what it was (hipMemGetInfo; the allocating entry point
        # grows the context's buffers on its first weighted call, which was made above)
        assert free_bytes() == free_before
        # and the other way round, ending on an unweighted call (accumulator only)
        for w, wb, ws in zip(reversed(WEIGHTS), reversed(want_big), reversed(want_small)):
            assert small.weight_cost_static(1, 0, *w) == ws, w
            assert big.weight_cost_static(1, 0, *w) == wb, w
    finally:
        big.close()
        small.close()


def test_static_scratch_calls_from_two_contexts_take_turns():
    """Two contexts on one GPU share the array; their calls, made from two host threads at once, hold the GPU's lock
    until the result is back, so each still gets the allocating entry point's numbers."""
    import threading
    big = api.Context(1280, 720, depth=8, num_slots=2)
    small = api.Context(416, 240, depth=8, num_slots=2)
    try:
        _prepare(big, 1280, 720, 8, 7)
        _prepare(small, 416, 240, 8, 8)
        want = {id(c): [c.weight_cost(1, 0, *w) for w in WEIGHTS] for c in (big, small)}
        bad = []

        def worker(c):
            try:
                for rep in range(25):
                    for w, expect in zip(WEIGHTS, want[id(c)]):
                        got = c.weight_cost_static(1, 0, *w)
                        if got != expect:
                            bad.append((rep, w, got, expect))
            except Exception as e:      # noqa: BLE001 - reported by the parent thread
                bad.append(e)
        threads = [threading.Thread(target=worker, args=(c,)) for c in (big, small)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not bad, bad[:4]
    finally:
        big.close()
        small.close()


def test_static_scratch_rejects_oversized_plane():
    """7680x4320 at 10 bit: the padded half-res plane (18.7 MB) is beyond the 8 MiB array."""
    ctx = api.Context(7680, 4320, depth=10, num_slots=2)
    try:
        # without a weighted plane only the accumulator is used: allowed at any size
        frame = np.full((4320, 7680), 512, np.uint16)
        ctx.lowres_init(0, frame, 0)
        ctx.lowres_init(1, frame, 1)
        ctx.intra_estimate([0, 1])
        assert ctx.weight_cost_static(1, 0) == ctx.weight_cost(1, 0)
        with pytest.raises(api.X265GpuError, match="does not fit"):
            ctx.weight_cost_static(1, 0, 1, 64, 6, 0)
    finally:
        ctx.close()
