#!/bin/bash
# Instruction-cache behaviour of the two neighbour kernels (one rocprofv3 --pmc pass, kernel trace only).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_icache
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $ROOT/gpurun_out/pmc_icache -- python3 $ROOT/tools/run_steps.py c2 300 20 > $ROOT/gpurun_out/pmc_icache.log 2>&1 || tail -3 $ROOT/gpurun_out/pmc_icache.log
for K in "k_neighbours2<false, 1>" "k_neighbours2<false, 2>"; do python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_icache "$K"; done
