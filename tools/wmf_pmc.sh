cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof5_wmf; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_IFETCH SQ_ACTIVE_INST_VALU --kernel-include-regex "wmf_" --output-format csv -d $O/sq -- python3 tools/wmf_only.py > $O/sq.log 2>&1; echo rc1=$?
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-include-regex "wmf_" --output-format csv -d $O/sqc -- python3 tools/wmf_only.py > $O/sqc.log 2>&1; echo rc2=$?
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY --kernel-include-regex "wmf_" --output-format csv -d $O/sq2 -- python3 tools/wmf_only.py > $O/sq2.log 2>&1; echo rc3=$?
tail -2 $O/sq.log | cut -c1-300
