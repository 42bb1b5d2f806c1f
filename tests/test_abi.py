"""The C-ABI library loads and exports every symbol include/pcfa_hip.h declares (no GPU needed)."""
import os
import re

import pytest

from tests.util import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "pcfa_hip.h")).read()
    return sorted(set(re.findall(r"PCFA_API[^;(]*?\b(pcfa_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ("pcfa_corr_pyramid_fwd", "pcfa_corr_pyramid_bwd", "pcfa_corr_lookup_fwd", "pcfa_corr_lookup_bwd",
                 "pcfa_spatial_corr_fwd", "pcfa_spatial_corr_bwd", "pcfa_flow_loss_fwd", "pcfa_flow_loss_bwd",
                 "pcfa_box_transform_fwd", "pcfa_box_transform_bwd"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from pcfa_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _hip.load()
    assert lib.pcfa_abi_version() == _hip.ABI_VERSION
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert set(_hip.SIGNATURES) == set(declared_symbols())


def test_host_only_entry_points():
    """Pure host helpers can be called without a GPU."""
    import ctypes
    from pcfa_amd import _hip
    lib = _hip.load()
    # 4x4-tiled levels padded to whole tiles: 56x128 + 28x64 + 16x32 + 8x16
    assert lib.pcfa_corr_slab_floats(55, 128, 4) == 7168 + 1792 + 512 + 128 + 16   # + one zero tile
    h, w = ctypes.c_int(), ctypes.c_int()
    assert lib.pcfa_corr_level_offset(55, 128, 4, 2, ctypes.byref(h), ctypes.byref(w)) == 7168 + 1792
    assert (h.value, w.value) == (13, 32)
    assert lib.pcfa_corr_slab_floats(17, 21, 4) % 16 == 0
    assert lib.pcfa_corr_tiled_index(55, 128, 4, 0, 0, 0) == 0
    assert lib.pcfa_corr_tiled_index(55, 128, 4, 0, 5, 6) == ((1 * 32 + 1) * 16 + 1 * 4 + 2)
    assert lib.pcfa_corr_tiled_index(55, 128, 4, 0, 55, 0) == -1
    from pcfa_amd import hip_ops
    for l, (idx, hl, wl) in enumerate(hip_ops.tiled_index_maps(17, 21, 4)):
        assert int(idx[-1]) == lib.pcfa_corr_tiled_index(17, 21, 4, l, hl - 1, wl - 1)
        assert int(idx[wl + 1]) == lib.pcfa_corr_tiled_index(17, 21, 4, l, 1, 1)
    oh, ow = ctypes.c_int(), ctypes.c_int()
    assert lib.pcfa_spatial_corr_out_size(11, 9, 3, 3, 1, 1, 1, 1, 2, 2, ctypes.byref(oh), ctypes.byref(ow)) == 0
    assert (oh.value, ow.value) == (6, 5)
    oc = ctypes.c_int()
    # FlowNetC's layer at 448x1024 / 8 (correlation_cuda.cc:25-35): 441 x 56 x 128
    assert lib.pcfa_flownet_corr_out_size(56, 128, 20, 1, 20, 1, 2, ctypes.byref(oc), ctypes.byref(oh),
                                          ctypes.byref(ow)) == 0
    assert (oc.value, oh.value, ow.value) == (441, 56, 128)
    assert lib.pcfa_flownet_corr_out_size(10, 12, 2, 1, 4, 1, 2, ctypes.byref(oc), ctypes.byref(oh),
                                          ctypes.byref(ow)) == 0
    assert (oc.value, oh.value, ow.value) == (25, 6, 8)
    assert lib.pcfa_flownet_corr_out_size(10, 12, 0, 2, 4, 1, 2, None, None, None) == -1  # even kernel_size
    assert lib.pcfa_status_string(-2).decode().startswith("unsupported")
    # argument validation happens before any launch
    assert lib.pcfa_corr_lookup_fwd(None, None, None, 1, 8, 8, 4, 4, None) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from pcfa_amd import hip_ops
    x = torch.zeros(1, 4, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_ops.CorrBlock(x, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_ops.spatial_correlation_sample(x, x, patch_size=9)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_ops.loss_delta_constraint(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), x, x)
