#!/bin/bash
# tools/launch_trace.sh [out_dir] [extra bench args] — per-launch durations of one view of the bench workload (rocprofv3 --kernel-trace):
# prints the 16 sweep launches of the LAST step in order with their kernel instantiation (rolled / packed form).
O=${1:-gpurun_out/trace}; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}" || exit 1
mkdir -p "$O"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-boundary --no-strict-record "$@" > "$O/bench.log" 2>&1 || exit 1
python3 - "$O" <<'PY'
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + "/kt/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "pm_sweep_kernel" in r["Kernel_Name"] or "pm_full_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-17:]
tot = 0.0
for r in last:
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += ms
    m = re.search(r"<(.*)>", r["Kernel_Name"])
    print(f"{ms:8.3f} ms  {m.group(1) if m else r['Kernel_Name'][:80]}")
print(f"{tot:8.3f} ms  total of the view's init + 16 sweeps")
PY
