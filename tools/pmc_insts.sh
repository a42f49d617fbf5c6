#!/bin/bash
# Instruction mix / wait counters of the two neighbour kernels (one rocprofv3 --pmc pass, kernel trace only).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_insts
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $ROOT/gpurun_out/pmc_insts -- python3 $ROOT/tools/run_steps.py c2 300 20 > $ROOT/gpurun_out/pmc_insts.log 2>&1 || tail -3 $ROOT/gpurun_out/pmc_insts.log
for K in "k_neighbours2<false, 1>" "k_neighbours2<false, 2>"; do python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_insts "$K"; done
