"""Regenerates tests/golden/pm_small.npz.

These are REGRESSION vectors produced by this repository's own CPU oracle (oracle/tsar_oracle.c), not
outputs of the reference: the reference ships no fixtures and cannot be built in this image.
They freeze the oracle's behaviour (so an accidental change of semantics shows up in the CPU suite) and
give the GPU suite a committed expected result that does not depend on the oracle being rebuilt.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as ol  # noqa: E402
from tsar_mvs_amd import synth  # noqa: E402


def main():
    sc = synth.make_scene(64, 48, 3, seed=99)
    imgs = [im.numpy() for im in sc.images]
    orc = ol.Oracle(imgs, sc.K, sc.R, sc.t, sc.depth_min, sc.depth_max, box=11, n_best=1, seed=4242)
    gt = synth.gt_planes(sc).numpy()
    cost_gt, bv_gt, rt_gt = orc.pm_cost_planes(gt)
    orc.pm_init()
    init_n, init_c = orc.norm4.copy(), orc.c.copy()
    orc.pm_iterate(1)
    it1_n, it1_c, it1_bv = orc.norm4.copy(), orc.c.copy(), orc.beview.copy()
    out = orc.compute_disp()
    np.savez_compressed(
        os.path.join(HERE, "pm_small.npz"),
        images=np.stack(imgs).astype(np.uint8), K=sc.K, R=sc.R, t=sc.t, depth_min=np.float32(sc.depth_min), depth_max=np.float32(sc.depth_max),
        box=11, n_best=1, seed=4242, gt_planes=gt, cost_gt=cost_gt, beview_gt=bv_gt,
        init_planes=init_n, init_cost=init_c, it1_planes=it1_n, it1_cost=it1_c, it1_beview=it1_bv, out4=out)
    print("wrote", os.path.join(HERE, "pm_small.npz"))


if __name__ == "__main__":
    main()
