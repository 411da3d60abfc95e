#!/bin/bash
# The measurements profiles/ and DESIGN.md quote, on one MI355X:  tools/gpu_measure.sh <tag>   (outputs: gpurun_out/<tag>_*)
# One script for every round: earlier rounds kept a copy per GPU session (gpu_r2a..q.sh); their extra steps are flags of bench.py now.
export TMPDIR=/tmp
O=gpurun_out
T=$1
mkdir -p $O
python bench.py > $O/${T}_bench_uniform.json 2> $O/${T}_bench_uniform.err
python bench.py --workload ragged > $O/${T}_bench_ragged.json 2> $O/${T}_bench_ragged.err
CMD="python bench.py --no-cpu-baseline --no-live-traffic"
rocprofv3 --kernel-trace --stats -d $O/${T}_trace -o bench -- $CMD > $O/${T}_trace.json 2> $O/${T}_trace.err
SHORT="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${T}_pmc_fetch -o pmc -- $SHORT > /dev/null 2> $O/${T}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${T}_pmc_write -o pmc -- $SHORT > /dev/null 2> $O/${T}_pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/${T}_pmc_sq -o pmc -- $SHORT > /dev/null 2> $O/${T}_pmc_sq.err
RCMD="python bench.py --workload ragged --no-cpu-baseline --no-live-traffic"
rocprofv3 --kernel-trace --stats -d $O/${T}_trace_ragged -o bench -- $RCMD > $O/${T}_trace_ragged.json 2> $O/${T}_trace_ragged.err
RSHORT="python bench.py --workload ragged --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${T}_pmc_fetch_ragged -o pmc -- $RSHORT > /dev/null 2> $O/${T}_pmc_fetch_ragged.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${T}_pmc_write_ragged -o pmc -- $RSHORT > /dev/null 2> $O/${T}_pmc_write_ragged.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $O/${T}_pmc_sq_ragged -o pmc -- $RSHORT > /dev/null 2> $O/${T}_pmc_sq_ragged.err
for f in $(find $O -name "*_results.db" -path "*${T}_*" | sort); do python tools/rocpd_summary.py $f; done > $O/${T}_summary.txt 2>&1
python tools/measure_misc.py > $O/${T}_misc.json 2> $O/${T}_misc.err
python tools/measure_cli.py > $O/${T}_cli.json 2> $O/${T}_cli.err
find $O -name "*.db" -path "*${T}_*" -size +20M -delete
echo done
