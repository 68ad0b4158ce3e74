#!/bin/bash
# round-4 GPU job 4: fast blend as (t00 + ax d1) + ay (d2 + ax d3): A/B against the previous library on one box, then the GPU suite
set -o pipefail
O=gpurun_out/r4j12; mkdir -p $O
B="--no-cpu-baseline --no-host-boundary --no-strict-record --steps 4 --warmup 1"
for r in 1 2; do
  TSAR_LIB=$PWD/tsar-mvs_amd/libtsar_hip_prev.so timeout -k 10 200 python3 bench.py $B > $O/prev_$r.json 2> $O/prev_$r.err || { echo prev failed; tail -3 $O/prev_$r.err; exit 1; }
  timeout -k 10 200 python3 bench.py $B > $O/new_$r.json 2> $O/new_$r.err || { echo new failed; tail -3 $O/new_$r.err; exit 1; }
done
python3 -c "
import json
for n in ('prev_1','new_1','prev_2','new_2'):
    d=json.load(open('$O/'+n+'.json')); print(n, round(d['value'],3), d['kernel_ms'])"
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "fast_exact or baseline_configs or edges or parity or golden" > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
exit $rc
