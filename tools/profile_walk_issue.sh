R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace -d $R/gpurun_out/prof_ic -o ic -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_ic.log 2>&1; echo ic $?
rocprofv3 --pmc SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INST_CYCLES_SALU --kernel-trace -d $R/gpurun_out/prof_if -o if -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_if.log 2>&1; echo if $?
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/prof_ix -o ix -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_ix.log 2>&1; echo ix $?
