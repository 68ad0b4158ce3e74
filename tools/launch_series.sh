#!/bin/bash
# tools/launch_series.sh — per-launch durations of pm_sweep over the views of one bench run (rocprofv3 kernel trace), for the
# environment given on the command line, e.g.   bash tools/launch_series.sh gpurun_out/ls0 TSAR_BUFFER_GATHER=0
O=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}" || exit 1
mkdir -p "$O"
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-boundary --no-strict-record > $O/bench.log 2>&1 || exit 1
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "pm_sweep_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
views = [d[i:i + 16] for i in range(0, len(d), 16)]
for v in views[1:]:
    print(" ".join(f"{x:.1f}" for x in v), "| sum", round(sum(v), 1))
PY
rm -f $O/*/*kernel_trace.csv
