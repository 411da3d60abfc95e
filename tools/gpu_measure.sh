#!/bin/bash
# The measurements profiles/ and DESIGN.md quote, on one MI355X:  tools/gpu_measure.sh <tag>   (outputs: gpurun_out/<tag>_*)
# One script for every round: earlier rounds kept a copy per GPU session (gpu_r2a..q.sh); their extra steps are flags of bench.py now.
export TMPDIR=/tmp
O=gpurun_out
T=$1
mkdir -p $O
# the bench lines: default (configs[2] + the `ragged` key), --workload ragged, and both sustained for >= 10 s
python bench.py > $O/${T}_bench_uniform.json 2> $O/${T}_bench_uniform.err
python bench.py --workload ragged > $O/${T}_bench_ragged.json 2> $O/${T}_bench_ragged.err
python bench.py --min-seconds 10 --no-cpu-baseline --no-live-traffic --no-ragged-extra > $O/${T}_bench_uniform_sustained.json 2> $O/${T}_sustained.err
python bench.py --workload ragged --min-seconds 10 --no-cpu-baseline --no-live-traffic > $O/${T}_bench_ragged_sustained.json 2>> $O/${T}_sustained.err
# kernel trace + PMC passes (counters in their own runs, --kernel-trace only)
CMD="python bench.py --min-seconds 0 --no-cpu-baseline --no-live-traffic --no-ragged-extra"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_trace -o bench -- $CMD > $O/${T}_trace.json 2> $O/${T}_trace.err
SHORT="python bench.py --min-seconds 0 --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-ragged-extra"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${T}_pmc_fetch -o pmc -- $SHORT > /dev/null 2> $O/${T}_pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${T}_pmc_write -o pmc -- $SHORT > /dev/null 2> $O/${T}_pmc_write.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/${T}_pmc_sq -o pmc -- $SHORT > /dev/null 2> $O/${T}_pmc_sq.err
# (the traces above run the library's automatic encoder = the product's: enc_site16_kernel.)  The opt-in 12-slot encoder: the same trace with it selected
M6A_ENCODER=fast timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_trace_fast -o bench -- $CMD > $O/${T}_trace_fast.json 2> $O/${T}_trace_fast.err
RCMD="python bench.py --workload ragged --min-seconds 0 --no-cpu-baseline --no-live-traffic"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_trace_ragged -o bench -- $RCMD > $O/${T}_trace_ragged.json 2> $O/${T}_trace_ragged.err
RSHORT="python bench.py --workload ragged --min-seconds 0 --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${T}_pmc_fetch_ragged -o pmc -- $RSHORT > /dev/null 2> $O/${T}_pmc_fetch_ragged.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${T}_pmc_write_ragged -o pmc -- $RSHORT > /dev/null 2> $O/${T}_pmc_write_ragged.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $O/${T}_pmc_sq_ragged -o pmc -- $RSHORT > /dev/null 2> $O/${T}_pmc_sq_ragged.err
for f in $(find $O -name "*_results.db" -path "*${T}_*" | sort); do python tools/rocpd_summary.py $f; done > $O/${T}_summary.txt 2>&1
# the first call, as a timeline (HIP API + kernels) and by phase
timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/${T}_tl -o tl -- python bench.py --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-live-traffic --no-ragged-extra > /dev/null 2> $O/${T}_tl.err
python tools/first_call_timeline.py $O/${T}_tl > $O/${T}_first_call_timeline.txt 2>&1
python tools/first_call_probe.py > $O/${T}_first_call.json 2> $O/${T}_first_call.err
# microbenchmarks behind the ceilings, the stream generator, the feed API from plain C
./tools/valu_rate_bench > $O/${T}_valu_rate.json 2>&1
RG_PROBE_COPIES=1 ./tools/rg_probe > $O/${T}_rg_probe.txt 2>&1
python tools/h2d_probe.py 8 > $O/${T}_h2d_probe.json 2> $O/${T}_h2d_probe.err
python tools/step_ab.py 20 3 > $O/${T}_step_by_encoder.json 2>> $O/${T}_h2d_probe.err
python tools/time_encoder.py > $O/${T}_encoder_kernels.txt 2>&1
python tools/stream_probe.py > $O/${T}_stream.json 2> $O/${T}_stream.err
{ ./tools/feed_probe 20000 16; ./tools/feed_probe 20000 1024; ./tools/feed_probe 1000000 16 20 20; ./tools/feed_probe 1000000 4096 20 20; } > $O/${T}_feed_probe.json 2>&1
python tools/measure_misc.py > $O/${T}_misc.json 2> $O/${T}_misc.err
python tools/measure_cli.py > $O/${T}_cli.json 2> $O/${T}_cli.err
bash tools/measure_cli_gpus.sh > $O/${T}_cli_gpus.json 2>> $O/${T}_cli.err
# where the encoder's SIMD cycles go (both kernels, four counter groups) and the A/B builds of tools/encoder_ab.py (if tools/ko/ travelled)
bash tools/encoder_floor.sh ${T} > /dev/null 2>&1
if ls tools/ko/libm6a_ab_*.so > /dev/null 2>&1; then python tools/encoder_ab.py 3 > $O/${T}_encoder_ab.json 2> $O/${T}_encoder_ab.err; fi
# what a VALU instruction costs next to f32 MFMAs by placement, and the tile timeline of every wave from the kernel's own clock (diagnostic builds in tools/ko/)
./tools/mfma_valu_mix > $O/${T}_mfma_valu_mix.json 2>&1
if ls tools/ko/libm6a_ab_stamps*.so > /dev/null 2>&1; then python tools/encoder_timeline.py > $O/${T}_encoder_timeline.json 2> $O/${T}_encoder_timeline.err; fi
# the rows either side of the path, on this box's host: rooflines, the dataprep run DESIGN.md quotes, the whole pipeline, the CLI's CSVs
python tools/host_rooflines.py 4.0 1400 > $O/${T}_host_rooflines.json 2> $O/${T}_host_rooflines.err
python tools/measure_dataprep.py ${DATAPREP_GB:-22} --single > $O/${T}_dataprep.json 2> $O/${T}_dataprep.err
python tools/measure_pipeline.py > $O/${T}_pipeline.json 2> $O/${T}_pipeline.err
python tools/cli_csv_vs_reference.py > $O/${T}_cli_csv_vs_reference.txt 2>&1
# HIP kernels against the REFERENCE's own read probabilities at full size, if tests/golden/_big travelled (.gpurunignore line taken out for this call)
if ls tests/golden/_big/configs2_*.npy > /dev/null 2>&1; then python tests/report_full_size_vs_reference.py > $O/${T}_full_size_vs_reference.json 2> $O/${T}_full_size.err; fi
if ls tests/golden/_big/configs4_*.npy > /dev/null 2>&1; then python tests/report_full_size_vs_reference.py --ragged > $O/${T}_full_size_vs_reference_ragged.json 2>> $O/${T}_full_size.err; fi
find $O -name "*.db" -path "*${T}_*" -size +20M -delete
find $O -path "*${T}_tl*" -size +10M -delete
echo done
