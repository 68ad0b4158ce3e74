#!/bin/bash
# tools/startup_breakdown.sh — where the time of ONE tsar_gipuma process per view goes (the reference's shell loop, scripts/courtyard.sh:29-48):
# loader + static initialisation (--help), then three full-size single-view runs with --timing (host-side steps and kernels).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python3 - <<'PY'
import subprocess, time
for i in range(3):
    t = time.perf_counter(); subprocess.run(["./tsar-mvs_amd/tsar_gipuma", "--help"], capture_output=True)
    print("spawn + loader + static initialisation + exit, no GPU call (--help): %.1f ms" % ((time.perf_counter() - t) * 1e3))
PY
python tools/bench_cli_single.py --timing --repeat 3
