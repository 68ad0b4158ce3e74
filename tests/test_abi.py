"""The C-ABI library loads without a GPU and exports every symbol include/tsar.h declares."""
import ctypes
import os
import re

import pytest

from tsar_mvs_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "tsar.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsar_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _declared() == sorted(api.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    if not os.path.exists(api.LIB_PATH):
        ge.build()
    lib = ctypes.CDLL(api.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.tsar_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.tsar_version()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(api.Camera) == 21 * 4
    assert ctypes.sizeof(api.Params) == 40
    assert ctypes.sizeof(api.SlicSettings) == 20
    assert ctypes.sizeof(api.KernelTiming) == 56
    p = api.default_params()
    assert (p.box_hsize, p.box_vsize, p.n_best, p.cost_comb) == (19, 19, 2, 1)   # reference algorithmparameters.h:21-52


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.TsarError):
        api.Matcher()
