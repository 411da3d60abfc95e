#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
CMD="python bench.py --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 -L > $O/r2l_counters.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/r2l_pmc1 -o pmc -- $CMD > /dev/null 2> $O/r2l_pmc1.err
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace -d $O/r2l_pmc2 -o pmc -- $CMD > /dev/null 2> $O/r2l_pmc2.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES SQ_IFETCH SQ_INST_LEVEL_VMEM --kernel-trace -d $O/r2l_pmc3 -o pmc -- $CMD > /dev/null 2> $O/r2l_pmc3.err
for f in $(find $O -name "*_results.db" -path "*r2l*"); do python tools/rocpd_summary.py $f; done > $O/r2l_summary.txt 2>&1
find $O -name "*.db" -path "*r2l*" -size +20M -delete
echo done
