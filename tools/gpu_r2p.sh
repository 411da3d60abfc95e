#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or infer or full_size or beyond" > $O/r2p_pytest.log 2>&1
grep -n "passed\|failed" $O/r2p_pytest.log | tail -2
python bench.py --workload ragged --no-cpu-baseline --no-live-traffic > $O/r2p_bench_ragged.json 2> $O/r2p_bench_ragged.err
CMD="python bench.py --workload ragged --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic"
rocprofv3 --kernel-trace --stats -d $O/r2p_trace -o bench -- $CMD > $O/r2p_trace.json 2> $O/r2p_trace.err
for f in $(find $O -name "*_results.db" -path "*r2p*"); do python tools/rocpd_summary.py $f; done > $O/r2p_summary.txt 2>&1
python tools/measure_misc.py > $O/r2p_misc.json 2> $O/r2p_misc.err
echo done
