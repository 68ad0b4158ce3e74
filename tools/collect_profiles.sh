#!/bin/bash
# tools/collect_profiles.sh — the rocprofv3 passes behind profiles/rNN (run on the GPU box, e.g. through
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh gpurun_out/prof'
# then copy the CSVs named in profiles/rNN/README.md).  Each --pmc pass is its own run (never combined with a trace
# domain), restricted to the sweep kernel (rocprofv3 crashes inside torch's kernels otherwise), <= 8 SQ counters per pass.
set -o pipefail
O=${1:-gpurun_out/prof}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}" || exit 1
mkdir -p "$O"
B="--no-cpu-baseline --no-host-boundary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 $B > $O/bench_stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
    --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --iters 3 $B > $O/bench_pmc_sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum \
    --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_tcp -- python3 bench.py --steps 1 --warmup 0 --iters 3 $B > $O/bench_pmc_tcp.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --iters 3 $B > $O/bench_pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --iters 3 $B > $O/bench_pmc_write.log 2>&1
rc=$?
rm -f $O/stats/*/*kernel_trace.csv          # tens of MB; the stats CSV is what gets committed
exit $rc
