#!/bin/bash
# round-4 GPU job 1: ceilings of the shipping MIX body, strip-width sweep, full GPU suite on the refactored library
set -o pipefail
O=gpurun_out/r4j1; mkdir -p $O
export TSAR_LIB=$PWD/tsar-mvs_amd/libtsar_hip_exp.so
timeout -k 10 300 python3 tools/ab_converged.py --variants 2228474,6422778,10617082 --rounds 6 > $O/ceilings.json 2> $O/ceilings.err || { echo ceilings failed; tail -5 $O/ceilings.err; exit 1; }
unset TSAR_LIB
cat $O/ceilings.json
timeout -k 10 400 python3 tools/ab_sweep.py --rounds 3 --variants 250+TSAR_STRIP=24,250+TSAR_STRIP=6,250+TSAR_STRIP=8,250+TSAR_STRIP=12,250+TSAR_STRIP=16,250+TSAR_STRIP=32,250+TSAR_BLOCK=128 > $O/strips.json 2> $O/strips.err || { echo strips failed; tail -5 $O/strips.err; exit 1; }
cat $O/strips.json
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
exit $rc
