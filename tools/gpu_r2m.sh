#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
CMD="python bench.py --workload ragged --steps 6 --warmup 2 --no-cpu-baseline"
python bench.py --workload ragged --steps 20 --warmup 3 --no-cpu-baseline > $O/r2m_bench_ragged.json 2> $O/r2m_bench_ragged.err
rocprofv3 --kernel-trace --stats -d $O/r2m_trace -o bench -- $CMD > $O/r2m_trace.json 2> $O/r2m_trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $O/r2m_pmc_sq -o pmc -- $CMD > /dev/null 2> $O/r2m_pmc_sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $O/r2m_pmc_sq2 -o pmc -- $CMD > /dev/null 2> $O/r2m_pmc_sq2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r2m_pmc_fetch -o pmc -- $CMD > /dev/null 2> $O/r2m_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r2m_pmc_write -o pmc -- $CMD > /dev/null 2> $O/r2m_pmc_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/r2m_pmc_tcc -o pmc -- $CMD > /dev/null 2> $O/r2m_pmc_tcc.err
for f in $(find $O -name "*_results.db" -path "*r2m*"); do python tools/rocpd_summary.py $f; done > $O/r2m_summary.txt 2>&1
find $O -name "*.db" -path "*r2m*" -size +20M -delete
echo done
