#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
( time python bench.py ) > $O/r2n_bench_uniform.json 2> $O/r2n_bench_uniform.err
( time python bench.py --workload ragged --no-cpu-baseline ) > $O/r2n_bench_ragged.json 2> $O/r2n_bench_ragged.err
python tools/measure_misc.py > $O/r2n_misc.json 2> $O/r2n_misc.err
echo done
