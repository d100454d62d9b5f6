#!/bin/bash
# PMC pass over lab/gemm_lab (SQ counters only; separate from any trace/stats run)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/prof
mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_gemm_sq -- $OLDPWD/lab/gemm_lab) > $O/pmc_gemm_sq.log 2>&1
echo "rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_gemm_sq2 -- $OLDPWD/lab/gemm_lab) > $O/pmc_gemm_sq2.log 2>&1
echo "rc=$?"
tail -3 $O/pmc_gemm_sq.log
