#!/bin/bash
# round-4 GPU job 2: full GPU suite with the new tests, the default bench line, then TCP/L2 + wait counters per strip width
set -o pipefail
O=gpurun_out/r4j2; mkdir -p $O
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; rc=$?
tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo bench failed; tail -5 $O/bench_default.err; exit 1; }
cat $O/bench_default.json
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}" || exit 1
P="--steps 1 --warmup 0 --iters 3 --no-cpu-baseline --no-host-boundary --no-strict-record"
for cfg in STRIP=24 STRIP=12 STRIP=8 STRIP=32 BLOCK=128; do
  export TSAR_$cfg
  timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum \
      --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_tcp_$cfg -- python3 bench.py $P > $O/pmc_tcp_$cfg.log 2>&1 || { echo pmc tcp $cfg failed; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU \
      --kernel-include-regex pm_sweep --output-format csv -d $O/pmc_wait_$cfg -- python3 bench.py $P > $O/pmc_wait_$cfg.log 2>&1 || { echo pmc wait $cfg failed; exit 1; }
  unset TSAR_${cfg%%=*}
  echo "pmc $cfg done"
done
