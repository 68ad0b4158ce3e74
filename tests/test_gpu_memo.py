"""The propagation memo and the packed form of the sweep kernel (pm_sweep_impl.h SweepMemo / CMP): fewer evaluations, the same bits.

Within one tsar_pm_iterate call a pixel remembers the eight candidates of its previous propagation launch; a candidate that is the
same neighbour with a plane unchanged since was scored at this pixel then and rejected (or taken and since improved on), the
pixel's cost never rises, so it would be rejected again (gipuma.cu:555) and is not scored.  From the launch TSAR_COMPACT_FROM on, a
wave packs the surviving (pixel, arm) pairs 64 per trip, whichever lanes' pixels they belong to.  Neither may change a bit of the
state: the reference evaluates every hypothesis, every time.  The oracle-based whole-run tests (test_gpu_fast_exact.py,
test_gpu_baseline_configs.py) run through tsar_pm_iterate and so through both; this file compares the library with itself —
memo on / memo off / the packed form from the first launch it can serve / one call per iteration (no memo across calls) — at sizes
and settings the oracle cannot reach in test time, and checks that the packed launches did run."""
import os

import numpy as np
import pytest

from tsar_mvs_amd import api, synth

pytestmark = pytest.mark.gpu


def _run(scene, env, iters, flags, n_best=1, one_call=True, timing=False, box=11):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = api.matcher_from_scene(scene, box=box, n_best=n_best, seed=77, flags=flags)      # the knobs are read once, by tsar_create
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if timing:
        m.enable_kernel_timing(True)
    m.pm_init()
    if one_call:
        m.pm_iterate(iters)
    else:
        for _ in range(iters):
            m.pm_iterate(1)
    state = m.get_plane()
    t = m.kernel_timing() if timing else None
    m.close()
    return state, t


def _same(a, b):
    for u, v in zip(a, b):
        assert np.array_equal(np.ascontiguousarray(u).view(np.uint32), np.ascontiguousarray(v).view(np.uint32))


@pytest.mark.parametrize("mode", ["fast", "strict"])
@pytest.mark.parametrize("shape", ["128-thread workgroups", "256-thread workgroups"])
def test_memo_and_packed_form_change_no_bit(mode, shape):
    if shape.startswith("128"):
        sc = synth.make_scene(333, 251, 4, seed=3)             # below SWEEP_SMALL_IMAGE_TILES; the last tiles are partial in x and in y
    else:
        sc = synth.make_scene(2050, 1590, 3, seed=3)           # 6500 tiles of 256 threads, the last column 2 pixels wide, the last row 6 high:
                                                               # lanes without a pixel score other lanes' pairs in the packed form
    flags = 0 if mode == "fast" else api.FLAG_STRICT_DIV
    iters = 6
    plain, _ = _run(sc, {"TSAR_MEMO": "0"}, iters, flags)
    default, t = _run(sc, {}, iters, flags, timing=True)
    _same(plain, default)
    assert t["pm_sweep"][0] == 2 * iters and t["pm_sweep_packed"][0] == 2 * iters - 6          # packed from launch 6 of the call on
    early, t = _run(sc, {"TSAR_COMPACT_FROM": "2"}, iters, flags, timing=True)                 # ... from the first launch a memo exists for
    _same(plain, early)
    assert t["pm_sweep_packed"][0] == 2 * iters - 2
    rolled, t = _run(sc, {"TSAR_COMPACT_FROM": "-1"}, iters, flags, timing=True)               # the memo applied lane by lane in the rolled loop
    _same(plain, rolled)
    assert "pm_sweep_packed" not in t
    per_call, _ = _run(sc, {}, iters, flags, one_call=False)                                   # a memo never outlives its call
    _same(plain, per_call)
    changed = (plain[1] != _run(sc, {"TSAR_MEMO": "0"}, iters - 1, flags)[0][1]).mean()
    assert changed > 0.02                                                                       # the last iteration still moved pixels: the run was not idle


def test_packed_form_with_two_best_views_and_a_view_subset():
    """n_best 2 (the reference binary's default) takes the best-two kernel; a subset changes which views a pixel's cost is over"""
    sc = synth.make_scene(640, 480, 6, seed=9)
    a, _ = _run(sc, {"TSAR_MEMO": "0"}, 5, 0, n_best=2)
    b, t = _run(sc, {"TSAR_COMPACT_FROM": "2"}, 5, 0, n_best=2, timing=True)
    _same(a, b)
    assert t["pm_sweep_packed"][0] == 8


@pytest.mark.parametrize("mode", ["fast", "strict"])
@pytest.mark.parametrize("box,n_best", [(19, 2), (7, 1), (15, 5)])
def test_general_window_kernels_packed_form(mode, box, n_best):
    """the general-window tap loop (pm_core_lut.h: the reference binary's default box 19 / n_best 2, a small window, five best views —
    the 32-entry selection) through the same memo and packed form"""
    sc = synth.make_scene(800, 608, 6, seed=11)
    flags = 0 if mode == "fast" else api.FLAG_STRICT_DIV
    a, _ = _run(sc, {"TSAR_MEMO": "0"}, 4, flags, n_best=n_best, box=box)
    b, t = _run(sc, {"TSAR_COMPACT_FROM": "2"}, 4, flags, n_best=n_best, box=box, timing=True)
    _same(a, b)
    assert t["pm_sweep_packed"][0] == 6


def test_a_changed_subset_voids_the_memo():
    """the stored costs belong to the previous subset after tsar_set_view_subset: no skipping until the state is consistent again
    (cost_consistent false -> memo off), and the result is what a context without memo returns"""
    sc = synth.make_scene(320, 240, 5, seed=4)
    outs = []
    for env in ({"TSAR_MEMO": "0"}, {"TSAR_COMPACT_FROM": "2"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            m = api.matcher_from_scene(sc, seed=5, subset=[1, 2])
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        m.pm_init(); m.pm_iterate(3)
        m.set_view_subset([3, 4, 1])
        m.pm_iterate(3)
        outs.append(m.get_plane())
        m.close()
    _same(outs[0], outs[1])


@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_bench_workload_full_size_memo_changes_no_bit(mode):
    """BASELINE configs[1] as bench.py runs it (6048 x 4032, 1 + 10 views, 8 iterations in one call): the state after the run with
    the memo and the packed launches equals the state without, all 24.4 M pixels, planes, costs, best views and ratios"""
    sc = synth.make_scene(6048, 4032, 10, device="cuda", seed=1234)
    flags = 0 if mode == "fast" else api.FLAG_STRICT_DIV
    a, _ = _run(sc, {"TSAR_MEMO": "0"}, 8, flags)
    b, t = _run(sc, {}, 8, flags, timing=True)
    _same(a, b)
    assert t["pm_sweep_packed"][0] == 10
