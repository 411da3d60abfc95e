#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r2i_pytest.log 2>&1
grep -n "passed\|failed" $O/r2i_pytest.log | tail -2
python bench.py --no-cpu-baseline > $O/r2i_bench_uniform.json 2> $O/r2i_bench_uniform.err
CMD="python bench.py --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/r2i_trace -o bench -- $CMD > $O/r2i_trace.json 2> $O/r2i_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r2i_pmc_fetch -o pmc -- $CMD > /dev/null 2> $O/r2i_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r2i_pmc_write -o pmc -- $CMD > /dev/null 2> $O/r2i_pmc_write.err
CMD="python bench.py --workload ragged --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/r2i_trace_ragged -o bench -- $CMD > $O/r2i_trace_ragged.json 2> $O/r2i_trace_ragged.err
for f in $(find $O -name "*_results.db" -path "*r2i*"); do python tools/rocpd_summary.py $f; done > $O/r2i_summary.txt 2>&1
find $O -name "*.db" -path "*r2i*" -size +20M -delete
echo done
