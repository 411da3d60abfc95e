#!/bin/bash
# Where the read encoder's time goes, on HEAD:  tools/encoder_floor.sh <tag>  -> gpurun_out/<tag>_encoder_floor.txt
# Both encoder kernels on the default bench workload (20 M reads = 625 000 tiles of 32 reads): the automatic 16-slot kernel
# (enc_site16_kernel) and the opt-in 12-slot one (M6A_ENCODER=fast: enc_csite_kernel).  Four SQ/GRBM counter groups, one rocprofv3
# run each, --kernel-trace only (MI355X_MICROARCH.md: 8 SQ slots per pass; never combined with other trace domains).
export TMPDIR=/tmp
O=gpurun_out
T=$1
mkdir -p $O
SHORT="python bench.py --min-seconds 0 --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-ragged-extra"
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
G2="SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INST_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
G4="GRBM_GUI_ACTIVE GRBM_COUNT"
for ENC in auto fast; do
  i=0
  for G in "$G1" "$G2" "$G3" "$G4"; do
    i=$((i+1))
    M6A_ENCODER=$ENC timeout 600 rocprofv3 --pmc $G --kernel-trace -d $O/${T}_floor_${ENC}_g$i -o pmc -- $SHORT > /dev/null 2> $O/${T}_floor_${ENC}_g$i.err
  done
done
python tools/encoder_floor.py $O ${T} > $O/${T}_encoder_floor.txt 2>&1
find $O -name "*.db" -path "*${T}_floor_*" -size +20M -delete
tail -40 $O/${T}_encoder_floor.txt
