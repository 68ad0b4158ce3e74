#!/bin/bash
# tools/collect_profiles_r05.sh — round 5's rocprofv3 passes (run on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles_r05.sh gpurun_out/prof5'
# then copy the CSVs named in profiles/r05/README.md).  Same rules as tools/collect_profiles.sh: each --pmc pass is its own run, never
# combined with a trace domain, <= 8 counters, restricted to the matcher's kernels.  New in round 5: the counters of the RANDOM-PLANE
# kernels — pm_full_kernel<INIT> (the initialisation) and the first sweep launch of a view — which rounds 1-4 only argued about
# ("six cache lines per lane, view and hypothesis"): the include regex takes pm_full_kernel as well.  The PMC passes run one whole view
# (8 iterations) since the memo / packed launches of the later iterations differ from the early ones.
set -o pipefail
O=${1:-gpurun_out/prof5}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}" || exit 1
mkdir -p "$O"
B="--no-cpu-baseline --no-host-boundary --no-strict-record"
P="--steps 1 --warmup 0 $B"
RX="pm_full_kernel|pm_sweep"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py > $O/bench_stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
    --kernel-include-regex "$RX" --output-format csv -d $O/pmc_sq -- python3 bench.py $P > $O/bench_pmc_sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_IFETCH \
    --kernel-include-regex "$RX" --output-format csv -d $O/pmc_stall -- python3 bench.py $P > $O/bench_pmc_stall.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum \
    --kernel-include-regex "$RX" --output-format csv -d $O/pmc_tcp -- python3 bench.py $P > $O/bench_pmc_tcp.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RX" --output-format csv -d $O/pmc_fetch -- python3 bench.py $P > $O/bench_pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RX" --output-format csv -d $O/pmc_write -- python3 bench.py $P > $O/bench_pmc_write.log 2>&1
rc=$?
for f in $O/stats/*/*kernel_trace.csv; do
    [ -f "$f" ] && { head -1 "$f"; grep -E "pm_sweep_kernel|pm_full_kernel|compute_disp|split_out4|build_quad" "$f"; } > "$O/stats/matcher_launches.csv" && rm -f "$f"
done
exit $rc
