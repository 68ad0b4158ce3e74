#!/bin/bash
# tools/collect_profiles.sh — the rocprofv3 passes behind profiles/rNN (run on the GPU box, e.g. through
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh gpurun_out/prof'
# then copy the CSVs named in profiles/rNN/README.md).  Each --pmc pass is its own run (never combined with a trace
# domain), restricted to the sweep kernel (rocprofv3 crashes inside torch's kernels otherwise), <= 8 SQ counters per pass.
# The stats pass is the DEFAULT bench command (the one the driver runs), so the kernel averages are those of the bench line.
set -o pipefail
O=${1:-gpurun_out/prof}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}" || exit 1
mkdir -p "$O"
B="--no-cpu-baseline --no-host-boundary --no-strict-record"
P="--steps 1 --warmup 0 --iters 3 $B"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py > $O/bench_stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
    --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_sq -- python3 bench.py $P > $O/bench_pmc_sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_IFETCH \
    --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_stall -- python3 bench.py $P > $O/bench_pmc_stall.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum \
    --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_tcp -- python3 bench.py $P > $O/bench_pmc_tcp.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_fetch -- python3 bench.py $P > $O/bench_pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_write -- python3 bench.py $P > $O/bench_pmc_write.log 2>&1 &&
# strict mode's sweeps (round 4: what bounds the oracle-exact kernel): the same two counter sets with the strict record left in
# (rows whose kernel name carries <..., true, true, 122 / 131194, ...> are the strict launches)
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
    --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_strict_sq -- python3 bench.py --steps 1 --warmup 0 --iters 3 --no-cpu-baseline --no-host-boundary > $O/bench_pmc_strict_sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum \
    --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_strict_tcp -- python3 bench.py --steps 1 --warmup 0 --iters 3 --no-cpu-baseline --no-host-boundary > $O/bench_pmc_strict_tcp.log 2>&1
rc=$?
# the per-dispatch trace is tens of MB (torch renders the synthetic scene); keep the matcher's own launches only
for f in $O/stats/*/*kernel_trace.csv; do
    [ -f "$f" ] && { head -1 "$f"; grep -E "pm_sweep_kernel|pm_full_kernel|compute_disp|split_out4|build_quad" "$f"; } > "$O/stats/matcher_launches.csv" && rm -f "$f"
done
exit $rc
