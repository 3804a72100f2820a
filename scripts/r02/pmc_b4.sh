#!/bin/bash
# SQ counters of the forward kernel selected by $1 (auto|mfma|b4), logp-only and full tapes; separate --pmc passes
set -u
KERN=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export KF_FORWARD_KERNEL=$KERN
for MODE in logp full; do
  EXTRA=""; [ "$MODE" = "logp" ] && EXTRA="--logp-only"
  scripts/pmc.sh r02_pmc_${KERN}_${MODE}_A "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" $EXTRA "$@" > $ROOT/gpurun_out/r02_pmc_${KERN}_${MODE}_A.txt 2>&1
  scripts/pmc.sh r02_pmc_${KERN}_${MODE}_B "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" $EXTRA "$@" > $ROOT/gpurun_out/r02_pmc_${KERN}_${MODE}_B.txt 2>&1
  scripts/pmc.sh r02_pmc_${KERN}_${MODE}_C "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR" $EXTRA "$@" > $ROOT/gpurun_out/r02_pmc_${KERN}_${MODE}_C.txt 2>&1
done
cat $ROOT/gpurun_out/r02_pmc_${KERN}_*.txt
