"""CPU: the C-ABI library loads and exports every symbol include/x265gpu.h declares.

No compute is called here (there is no GPU in the build container); argument
validation and the no-device error path are exercised instead.
"""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "x265gpu.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(x265gpu_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from x265_amd import api
    if not os.path.exists(api.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return api.load_library()


def test_header_and_binding_agree(lib):
    from x265_amd import api
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(api.EXPORTS) == names
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n


def test_header_cites_reference_for_every_entry_point():
    text = open(HEADER).read()
    # every group of declarations carries a file:line citation of what it replaces
    assert len(re.findall(r"(slicetype|lowres|bitcost|picyuv|primitives|common)\.(cpp|h):\d+", text)) >= 15


def test_open_rejects_bad_arguments(lib):
    from x265_amd import api
    h = C.c_void_p()
    assert lib.x265gpu_open(None, C.byref(h)) == -1
    cfg = api.Config()
    cfg.sourceWidth, cfg.sourceHeight, cfg.internalBitDepth = 8, 8, 8
    cfg.maxCUSize, cfg.bframes, cfg.numSlots = 64, 4, 1
    cfg.searchMethod, cfg.merange = api.HEX_SEARCH, 16
    assert lib.x265gpu_open(C.byref(cfg), C.byref(h)) == -1          # too small
    cfg.sourceWidth, cfg.sourceHeight = 320, 192
    cfg.internalBitDepth = 9
    assert lib.x265gpu_open(C.byref(cfg), C.byref(h)) == -1          # no such depth
    cfg.internalBitDepth = 8
    cfg.bframes = 17
    assert lib.x265gpu_open(C.byref(cfg), C.byref(h)) == -1          # beyond X265_BFRAME_MAX
    cfg.bframes = 4
    assert not h.value


def test_no_cpu_fallback(lib):
    """Without a GPU the product path must fail loudly, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from x265_amd import api
    with pytest.raises(api.X265GpuError):
        api.Context(320, 192)


def test_null_context_is_an_error(lib):
    assert lib.x265gpu_sync(None) == -1
    assert lib.x265gpu_estimate(None, 0, None) == -1
    assert lib.x265gpu_get_geometry(None, None) == -1
    lib.x265gpu_close(None)


def test_host_layer_exports():
    """The C++ host layer's plain-C face (x265_amd/host/*.h) loads beside the library."""
    from x265_amd import api
    if not os.path.exists(api.HOST_LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    H = api.load_host_library()
    text = "".join(open(os.path.join(ROOT, "x265_amd", "host", f)).read()
                   for f in ("lookahead_host.h", "threaded_me_host.h"))
    declared = sorted(set(re.findall(r"\b(xgh_[a-z_]+)\s*\(", text)))
    assert declared == sorted(api.HOST_EXPORTS)
    for n in declared:
        assert hasattr(H, n), "missing export %s" % n
    assert H.xgh_window_create(None, 4, 0, 8) is None
