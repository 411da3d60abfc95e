#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or infer or smoke or full_size or device_tensors or job_offset or soak" > $O/r2b_pytest.log 2>&1
tail -5 $O/r2b_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r2b_smoke.log 2>&1; tail -2 $O/r2b_smoke.log
python bench.py --workload ragged --steps 20 --warmup 3 --no-cpu-baseline > $O/r2b_bench_ragged.json 2> $O/r2b_bench_ragged.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r2b_bench_uniform.json 2> $O/r2b_bench_uniform.err
python tools/measure_misc.py > $O/r2b_misc.json 2> $O/r2b_misc.err
echo done
