#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r2e_pytest.log 2>&1
tail -4 $O/r2e_pytest.log
python tools/host_path_sweep.py > $O/r2e_host_sweep.json 2> $O/r2e_host_sweep.err
python tools/measure_cli.py > $O/r2e_cli.json 2> $O/r2e_cli.err
M6A_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --sites 200000 --steps 5 --warmup 2 > $O/r2e_bench_gloo2.json 2> $O/r2e_bench_gloo2.err
echo done
