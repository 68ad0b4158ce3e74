#!/usr/bin/env python3
"""tools/sweep_repeat_census.py — the census behind the propagation memo: per half-sweep of the bench workload, the share of alive arms
whose candidate plane the pixel already tried in the previous iteration (tsar_selftest_sweep_repeat), what the rolled loop could skip
wave-uniformly (nothing) and what lane-local queues would (max over lanes): why the packed form exists."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tsar_mvs_amd import api, synth  # noqa: E402
W, H, V, IT = 6048, 4032, 10, 8
sc = synth.make_scene(W, H, V, device=torch.device("cuda", 0), seed=1234, cam_seed=42, step=0.03)
m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024)
L = m.L
memo = torch.zeros((H * W * 8,), dtype=torch.int64, device="cuda")
m.pm_init()
for it in range(IT):
    for colour in (0, 1):
        out = (C.c_uint64 * 8)()
        assert L.tsar_selftest_sweep_repeat(m._ctx, colour, C.c_void_p(memo.data_ptr()), out) == 0
        m.pm_sweep(colour)
        o = list(out)
        print(json.dumps({"iter": it, "colour": colour, "alive_arms": o[0], "repeat_same_arm": round(o[1] / max(o[0], 1), 4), "repeat_any_arm": round(o[2] / max(o[0], 1), 4),
                          "wave_arm_pairs": o[3], "pairs_all_lanes_repeat": round(o[4] / max(o[3], 1), 4),
                          "lane_queue_now_per_wave": round(o[5] / (o[3] / 8.0), 3) if o[3] else None, "lane_queue_fresh_per_wave": round(o[6] / (o[3] / 8.0), 3) if o[3] else None,
                          "packed_trips_per_wave": round(o[7] / (o[3] / 8.0), 3) if o[3] else None}), flush=True)
m.close()
