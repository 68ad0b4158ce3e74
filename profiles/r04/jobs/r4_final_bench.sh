#!/bin/bash
# round-4 final bench records (the final library): default line, the driver's command, the other configurations, stages, CLI
set -o pipefail
O=gpurun_out/r4final; mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo default failed; tail -3 $O/bench_default.err; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_20steps.json 2> $O/driver.err || { echo driver cmd failed; exit 1; }
timeout -k 10 200 python3 bench.py --width 640 --height 480 --views 4 --no-cpu-baseline --no-host-boundary > $O/bench_cfg1_640x480.json 2>/dev/null || { echo cfg1 failed; exit 1; }
timeout -k 10 300 python3 bench.py --width 3840 --height 2160 --views 20 --iters 12 --steps 3 --warmup 1 --no-cpu-baseline --no-host-boundary > $O/bench_cfg5_3840x2160_20views_12iters.json 2>/dev/null || { echo cfg5 failed; exit 1; }
timeout -k 10 300 python3 bench.py --box 19 --n_best 2 --steps 2 --warmup 1 --no-cpu-baseline --no-host-boundary > $O/bench_box19_nbest2.json 2>/dev/null || { echo box19 failed; exit 1; }
timeout -k 10 300 python3 tools/bench_stages.py > $O/stages_cfg4.txt 2> $O/stages.err || { echo stages failed; tail -3 $O/stages.err; exit 1; }
python3 -c "
import json
for n in ('bench_default','bench_driver_command_20steps','bench_cfg1_640x480','bench_cfg5_3840x2160_20views_12iters','bench_box19_nbest2'):
    d=json.load(open('$O/'+n+'.json')); print(n, round(d['value'],2), 'strict', round(d.get('strict',{}).get('value',0),2), d['kernel_ms'].get('pm_sweep'), d['config'].get('tolerance',{}) and d['config']['tolerance'].get('depth_within_1e-3'))"
tail -25 $O/stages_cfg4.txt | cut -c1-200
