"""The C-ABI library loads and exports what include/ndt_hip.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from ndt_amd import hip as nh
from ndt_amd import flat_scene as fsmod


def header_text():
    with open(os.path.join(ROOT, "include", "ndt_hip.h")) as f:
        return f.read()


def test_header_declares_what_the_binding_expects():
    declared = set(re.findall(r"\b(ndt_hip_[a-z_0-9]+)\s*\(", header_text()))
    assert declared == set(nh.API_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = nh.load_library()
    for name in nh.API_SYMBOLS:
        assert hasattr(lib, name), name
    assert lib.ndt_hip_abi_version() == fsmod.ABI_VERSION


def test_struct_sizes_match_header_layout():
    # natural alignment of the C structs in include/ndt_hip.h
    assert C.sizeof(fsmod.FlatLight) == 16 + 5 * 8
    assert C.sizeof(fsmod.FlatObject) == 14 * 4 + 8 * 8
    assert C.sizeof(fsmod.FlatKdNode) == 32
    assert C.sizeof(fsmod.RenderParams) == 16 * 4
    assert C.sizeof(fsmod.RenderStats) == 4 * 8 + 2 * 4 + 3 * 8 + 2 * 8
    # the library and the binding agree on the ABI revision the flat scene carries
    assert nh.load_library().ndt_hip_abi_version() == fsmod.ABI_VERSION == 3


def test_shard_rows_helper_agrees_with_library():
    lib = nh.load_library()
    for h in (1, 7, 72, 1080):
        for step in (1, 2, 3, 8):
            for begin in range(step):
                assert lib.ndt_hip_shard_rows(h, begin, step) == fsmod.shard_rows(h, begin, step)
                assert fsmod.shard_rows(h, begin, step) == len(range(begin, h, step))


def test_no_device_is_a_loud_error_not_a_fallback():
    """Without a GPU the library must refuse, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nh.NdtHipError) as e:
        nh.NdtHip(0)
    assert e.value.code == fsmod.NDT_E_DEVICE
