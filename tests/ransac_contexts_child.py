"""Child process of test_ransac_two_contexts_share_the_device / test_ransac_two_processes_share_the_device (test_gpu_parity.py).

Runs in a FRESH interpreter (python -X faulthandler) so that whatever happens at interpreter / runtime teardown — the place the
round-3 core dumps came from, after pytest had already reported "passed" — ends up in THIS process's exit status and stderr, where
the parent test asserts on it.  N host threads, one context each, all on device 0, every thread fitting the same regions `reps`
times (ctypes releases the GIL: the contexts' kernels and cooperative launches overlap).  Inputs (the oracle's converged state,
reliability mask, regions) come from the parent as an .npz; outputs go back the same way.  No oracle in here: the parent checks.

    python -X faulthandler tests/ransac_contexts_child.py state.npz out.npz --threads 2 --reps 6
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("state")
ap.add_argument("out")
ap.add_argument("--threads", type=int, default=2)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--start-at", type=float, default=0.0, help="wall-clock time (time.time()) at which the fits start: lines two processes up")
a = ap.parse_args()

from tsar_mvs_amd import api, synth  # noqa: E402

st = np.load(a.state)
sc = synth.make_scene(int(st["w"]), int(st["h"]), int(st["n_src"]), seed=int(st["scene_seed"]))
ms = []
for _ in range(a.threads):
    m = api.matcher_from_scene(sc, seed=int(st["seed"]), flags=api.FLAG_STRICT_DIV)
    m.set_plane(st["norm4"].copy(), st["c"].copy())
    m.getview()
    m.set_reliable_mask(st["scale"])
    m.set_regions(st["labels"], st["text"], st["size"])
    ms.append(m)
out = [[] for _ in range(a.threads)]


def work(k):
    for _ in range(a.reps):
        out[k].append(ms[k].ransac_regions())


while time.time() < a.start_at:
    time.sleep(0.001)
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k,)) for k in range(a.threads)]
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.perf_counter() - t0
for m in ms:
    m.close()
np.savez(a.out, planes=np.stack([np.stack([p for p, _ in o]) for o in out]), ratio=np.stack([np.stack([r for _, r in o]) for o in out]), seconds=dt)
print("child done: %d threads x %d fits in %.3f s" % (a.threads, a.reps, dt), flush=True)
