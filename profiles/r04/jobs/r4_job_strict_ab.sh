#!/bin/bash
# strict-mode A/B on one box: previous library / new library, alternated; then the strict parity tests
set -o pipefail
O=gpurun_out/r4j8; mkdir -p $O
B="--no-cpu-baseline --no-host-boundary --steps 3 --warmup 1"
for r in 1 2; do
  TSAR_LIB=$PWD/tsar-mvs_amd/libtsar_hip_prev.so timeout -k 10 200 python3 bench.py $B > $O/prev_$r.json 2> $O/prev_$r.err || { echo prev failed; tail -3 $O/prev_$r.err; exit 1; }
  timeout -k 10 200 python3 bench.py $B > $O/new_$r.json 2> $O/new_$r.err || { echo new failed; tail -3 $O/new_$r.err; exit 1; }
done
python3 -c "
import json
for n in ('prev_1','new_1','prev_2','new_2'):
    d=json.load(open('$O/'+n+'.json')); print(n, 'fast', round(d['value'],3), 'strict', round(d['strict']['value'],3), round(d['strict']['pm_sweep_avg_launch_ms'],3))"
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q -k "baseline_configs or edges or parity" > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
exit $rc
