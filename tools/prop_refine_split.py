import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tsar_mvs_amd import api, synth
sc = synth.make_scene(6048, 4032, 10, device=torch.device("cuda", 0), seed=1234, cam_seed=42, step=0.03)
m = api.matcher_from_scene(sc, box=11, n_best=1, seed=2024)
m.pm_init(); m.pm_iterate(4)
m.enable_kernel_timing(True)
res = {}
for name, (dp, dr) in {"both": (1, 1), "prop_only": (1, 0), "refine_only": (0, 1), "neither": (0, 0)}.items():
    m.reset_kernel_timing()
    for rep in range(2):
        for colour in (0, 1):
            m.pm_sweep(colour, dp, dr)
    t = m.kernel_timing()["pm_sweep"]
    res[name] = round(t[1] / t[0], 3)
print(json.dumps(res))
m.close()
