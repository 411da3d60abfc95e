#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r2d_pytest.log 2>&1
tail -4 $O/r2d_pytest.log
python tools/measure_misc.py > $O/r2d_misc.json 2> $O/r2d_misc.err
python tools/measure_cli.py > $O/r2d_cli.json 2> $O/r2d_cli.err
python bench.py --workload ragged --steps 20 --warmup 3 --no-cpu-baseline > $O/r2d_bench_ragged.json 2> $O/r2d_bench_ragged.err
python bench.py --no-cpu-baseline > $O/r2d_bench_uniform.json 2> $O/r2d_bench_uniform.err
echo done
