cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_poolreg
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python tools/pool_reg_knockouts.py --one > $O/trace.out 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace -d $O/pmc1 -o p -- python tools/pool_reg_knockouts.py --one > $O/pmc1.out 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/pmc2 -o p -- python tools/pool_reg_knockouts.py --one > $O/pmc2.out 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace -d $O/pmc_vrb -o p -- tools/valu_rate_bench 4000 > $O/pmc_vrb.out 2>&1
for f in $(find $O -name "*_results.db" | sort); do python tools/rocpd_summary.py $f; done > $O/summary.txt 2>&1
find $O -name "*.db" -size +20M -delete
grep -E "pool_reg|klike|k_pkmul_dep2|kernel  |counter" $O/summary.txt | head -80
