#!/bin/bash
# round-4 GPU job 3: strict mode — sparse-K homography + column-major textures for the random-plane launches
set -o pipefail
O=gpurun_out/r4j3; mkdir -p $O
timeout -k 10 400 python3 tools/ab_sweep.py --strict --rounds 3 --variants 250+TSAR_STRICT_TRANSPOSED=0,250+TSAR_STRICT_TRANSPOSED=1,250+TSAR_STRICT_TRANSPOSED=2,250+TSAR_STRICT_TRANSPOSED=4 > $O/strict_tr.json 2> $O/strict_tr.err || { echo strict ab failed; tail -5 $O/strict_tr.err; exit 1; }
python3 -c "
import json;d=json.load(open('$O/strict_tr.json'))
for k,v in d.items(): print(k, 'sweep', round(v['sweep_ms_median'],2), 'init', round(v['init_ms_median'],2), 'gt', v['gt_1pct'])"
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-host-boundary > $O/bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('$O/bench.json')); print('fast', d['value'], 'strict', d['strict']['value'], d['strict'].get('pm_sweep_avg_launch_ms'), d['kernel_ms'])"
