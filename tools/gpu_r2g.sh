#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or infer or full_size or cli_drop or host_pointer" > $O/r2g_pytest.log 2>&1
tail -4 $O/r2g_pytest.log
python bench.py --workload ragged --steps 20 --warmup 3 --no-cpu-baseline > $O/r2g_bench_ragged.json 2> $O/r2g_bench_ragged.err
CMD="python bench.py --workload ragged --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/r2g_trace -o bench -- $CMD > $O/r2g_trace.json 2> $O/r2g_trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $O/r2g_pmc_sq -o pmc -- $CMD > /dev/null 2> $O/r2g_pmc_sq.err
for f in $(find $O -name "*_results.db" -path "*r2g*"); do python tools/rocpd_summary.py $f; done > $O/r2g_summary.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r2g_bench_uniform.json 2> $O/r2g_bench_uniform.err
find $O -name "*.db" -path "*r2g*" -size +20M -delete
echo done
