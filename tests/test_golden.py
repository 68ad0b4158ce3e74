"""Committed regression vectors (tests/golden/pm_small.npz, produced by tests/golden/make_golden.py from the
CPU oracle — NOT reference outputs, see that script's header): the oracle must keep reproducing them
(CPU suite) and the HIP path must reproduce them bit for bit in strict mode (GPU suite)."""
import os

import numpy as np
import pytest

import oracle_lib as ol

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pm_small.npz"))


def _inputs():
    imgs = [G["images"][k].astype(np.float32) for k in range(G["images"].shape[0])]
    return imgs, G["K"], G["R"], G["t"], float(G["depth_min"]), float(G["depth_max"])


def test_oracle_reproduces_golden():
    imgs, K, R, t, dmin, dmax = _inputs()
    o = ol.Oracle(imgs, K, R, t, dmin, dmax, box=int(G["box"]), n_best=int(G["n_best"]), seed=int(G["seed"]))
    c, bv, _ = o.pm_cost_planes(G["gt_planes"])
    assert np.array_equal(c, G["cost_gt"]) and np.array_equal(bv, G["beview_gt"])
    o.pm_init()
    assert np.array_equal(o.norm4.view(np.uint32), G["init_planes"].view(np.uint32)) and np.array_equal(o.c, G["init_cost"])
    o.pm_iterate(1)
    assert np.array_equal(o.norm4.view(np.uint32), G["it1_planes"].view(np.uint32))
    assert np.array_equal(o.c, G["it1_cost"]) and np.array_equal(o.beview, G["it1_beview"])
    assert np.array_equal(o.compute_disp(), G["out4"])


@pytest.mark.gpu
def test_hip_reproduces_golden():
    from tsar_mvs_amd import api
    imgs, K, R, t, dmin, dmax = _inputs()
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=int(G["box"]), box_vsize=int(G["box"]), n_best=int(G["n_best"]), depth_min=dmin, depth_max=dmax,
                                    seed=int(G["seed"]), flags=api.FLAG_STRICT_DIV))
    m.set_views(imgs, K, R, t)
    c, bv, _ = m.pm_cost_planes(G["gt_planes"])
    assert np.array_equal(c, G["cost_gt"]) and np.array_equal(bv, G["beview_gt"])
    m.pm_init()
    planes, cost, _, _ = m.get_plane()
    assert np.array_equal(planes.view(np.uint32), G["init_planes"].view(np.uint32)) and np.array_equal(cost, G["init_cost"])
    m.pm_iterate(1)
    planes, cost, bv, _ = m.get_plane()
    assert np.array_equal(planes.view(np.uint32), G["it1_planes"].view(np.uint32))
    assert np.array_equal(cost, G["it1_cost"]) and np.array_equal(bv, G["it1_beview"])
    m.compute_disp()
    res = m.get_result()
    assert np.array_equal(res["depth"], G["out4"][..., 3]) and np.array_equal(res["normal"], G["out4"][..., :3])
    m.close()
