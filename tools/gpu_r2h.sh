#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r2h_pytest.log 2>&1
tail -4 $O/r2h_pytest.log
CMD="python bench.py --workload ragged --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/r2h_trace -o bench -- $CMD > $O/r2h_trace.json 2> $O/r2h_trace.err
for f in $(find $O -name "*_results.db" -path "*r2h*"); do python tools/rocpd_summary.py $f; done > $O/r2h_summary.txt 2>&1
python bench.py --workload ragged --steps 20 --warmup 3 --no-cpu-baseline > $O/r2h_bench_ragged.json 2> $O/r2h_bench_ragged.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r2h_bench_uniform.json 2> $O/r2h_bench_uniform.err
find $O -name "*.db" -path "*r2h*" -size +20M -delete
echo done
