#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r2f_pytest.log 2>&1
tail -4 $O/r2f_pytest.log
python bench.py --workload ragged --steps 20 --warmup 3 --no-cpu-baseline > $O/r2f_bench_ragged.json 2> $O/r2f_bench_ragged.err
CMD="python bench.py --workload ragged --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/r2f_trace -o bench -- $CMD > $O/r2f_trace.json 2> $O/r2f_trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $O/r2f_pmc_sq -o pmc -- $CMD > /dev/null 2> $O/r2f_pmc_sq.err
for f in $(find $O -name "*_results.db" -path "*r2f*"); do python tools/rocpd_summary.py $f; done > $O/r2f_summary.txt 2>&1
python tools/measure_misc.py > $O/r2f_misc.json 2> $O/r2f_misc.err
find $O -name "*.db" -path "*r2f*" -size +20M -delete
echo done
