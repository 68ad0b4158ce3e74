#!/bin/bash
# tools/final_records_r05.sh [out_dir] — the bench lines and CLI figures of profiles/r05 on the library as it is (run on the GPU box)
O=${1:-gpurun_out/r5rec}
mkdir -p "$O"
set -o pipefail
python3 bench.py > "$O/bench_default.json" 2> "$O/e0.log" &&
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_command_20steps.json" 2> "$O/e1.log" &&
python3 bench.py --width 640 --height 480 --views 4 --steps 50 --warmup 5 --no-cpu-baseline --no-host-boundary > "$O/bench_cfg1_640x480.json" 2> "$O/e2.log" &&
python3 bench.py --width 3840 --height 2160 --views 20 --iters 12 --steps 3 --warmup 1 --no-cpu-baseline --no-host-boundary > "$O/bench_cfg5_3840x2160_20views_12iters.json" 2> "$O/e3.log" &&
python3 bench.py --box 19 --n_best 2 --steps 3 --warmup 1 --no-cpu-baseline --no-host-boundary > "$O/bench_box19_nbest2.json" 2> "$O/e4.log" &&
python3 tools/bench_cli.py --workers 1 > "$O/cli_all_8views.txt" 2>&1 &&
python3 tools/bench_cli_single.py > "$O/cli_single_view.txt" 2>&1 &&
python3 tools/sweep_repeat_census.py > "$O/sweep_repeat_census.jsonl" 2> "$O/e5.log" &&
bash tools/launch_trace.sh "$O/trace_fast" > "$O/launch_series_fast.txt" 2>&1 &&
python3 tools/prop_refine_split.py > "$O/prop_refine_split.json" 2> "$O/e6.log"
echo "rc=$?"
