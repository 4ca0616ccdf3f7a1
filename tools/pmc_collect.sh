#!/bin/bash
# Run on the GPU box (through gpurun): SQ counter passes over tools/pmc_driver.py (counters in their own runs, --kernel-trace only),
# then one table per kernel.   gpurun --timeout 1500 -- 'bash tools/pmc_collect.sh r02'
set -u
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  out=gpurun_out/pmc_${tag}_$i
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -o p -- python3 tools/pmc_driver.py > "$out/stdout.txt" 2> "$out/stderr.txt"
  echo "pass $i rc=$?"
  find "$out" -name "*kernel_trace.csv" -delete
done
python3 tools/pmc_table.py gpurun_out/pmc_${tag}_ > gpurun_out/${tag}_mfma_counters.csv
head -c 3000 gpurun_out/${tag}_mfma_counters.csv
