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


def test_chunk_length_of_the_general_window_loop_pads_least():
    """host logic of the general-window tap loop (pm_sweep_lut.hip lut_chunk_taps): a line of T taps is walked in chunks of 4, 5
    or 6; the chosen length pads the line least, the longer chunk on a tie.  No GPU needed: the chooser is host code."""
    import ctypes
    lib = ctypes.CDLL(api.LIB_PATH)
    f = getattr(lib, "_Z14lut_chunk_tapsi")
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int]
    for taps in range(1, 33):
        ch = f(taps)
        assert ch in (4, 5, 6)
        waste = {c: (c - taps % c) % c for c in (4, 5, 6)}
        assert waste[ch] == min(waste.values()), (taps, ch, waste)
        assert all(c <= ch for c in (4, 5, 6) if waste[c] == waste[ch]), (taps, ch, waste)
    assert f(6) == 6 and f(10) == 5 and f(4) == 4 and f(12) == 6 and f(8) == 4      # boxes 11, 19, 7, 23, 15
