"""The C-ABI library builds, loads and exports every symbol include/*.h declares (no GPU)."""
import ctypes
import glob
import os
import re

from conftest import ROOT


def declared_symbols():
    names = set()
    for header in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(header).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(?:int|long long)\s+(dfine_\w+)\s*\(", text))
    return names


def test_header_declares_symbols():
    assert "dfine_msda_fwd" in declared_symbols()


def test_library_exports_every_declared_symbol(hip_lib):
    from dfine_seg_amd import _lib

    declared = declared_symbols()
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in include/ but not exported"
    # the ctypes table must bind exactly the declared ABI
    assert set(_lib.SIGNATURES) == declared


def test_abi_version(hip_lib):
    assert hip_lib.dfine_abi_version() == 24


def test_bad_arguments_are_rejected_without_launch(hip_lib):
    # null pointers -> DFINE_E_BADARG (-1); no GPU is touched
    hw = (ctypes.c_int32 * 2)(4, 4)
    npts = (ctypes.c_int32 * 1)(2)
    rc = hip_lib.dfine_msda_fwd(None, 0, None, None, None, 1, 16, 1, 1, 32, 1, hw, npts, None)
    assert rc == -1


def test_cpu_tensors_fail_loudly(hip_lib):
    import pytest
    import torch

    from dfine_seg_amd import _lib, ops

    v = torch.zeros(1, 16, 1, 32)
    loc = torch.zeros(1, 1, 1, 2, 2)
    attn = torch.zeros(1, 1, 1, 2)
    with pytest.raises(_lib.DfineHipError):
        ops.deformable_attention_core(v, [(4, 4)], loc, attn, [2])
