#!/bin/bash
# round 2, GPU run A: probes + new bench lines + profile of the round-1 ragged kernels
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
./tools/rg_probe > $O/r2a_rg_probe.txt 2>&1
python bench.py > $O/r2a_bench_uniform.json 2> $O/r2a_bench_uniform.err
python bench.py --workload ragged --steps 20 --warmup 3 > $O/r2a_bench_ragged.json 2> $O/r2a_bench_ragged.err
M6A_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --sites 200000 --steps 5 --warmup 2 > $O/r2a_bench_gloo2.json 2> $O/r2a_bench_gloo2.err
for drv in 1 2; do
  CMD="python bench.py --workload ragged --scan-driver $drv --steps 6 --warmup 2 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats -d $O/r2a_trace_d$drv -o bench -- $CMD > $O/r2a_trace_d$drv.json 2> $O/r2a_trace_d$drv.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $O/r2a_pmc_sq_d$drv -o pmc -- $CMD > /dev/null 2> $O/r2a_pmc_sq_d$drv.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/r2a_pmc_fetch_d$drv -o pmc -- $CMD > /dev/null 2> $O/r2a_pmc_fetch_d$drv.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/r2a_pmc_write_d$drv -o pmc -- $CMD > /dev/null 2> $O/r2a_pmc_write_d$drv.err
done
for f in $(find $O -name "*_results.db" -path "*r2a*"); do python tools/rocpd_summary.py $f; done > $O/r2a_summary.txt 2>&1
find $O -name "*.db" -path "*r2a*" -size +20M -delete
echo done
