#!/bin/bash
# SQ counter passes (4 counters per pass, counters only: no trace domains) over tools/pmc_probe2.py <what>; results under
# gpurun_out/pmc_sq_<what>/<pass>/ and a per-kernel table on stdout (tools/pmc_sq_table.py).
what=${1:-gemm}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
out=gpurun_out/pmc_sq_$what
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python tools/pmc_probe2.py $what > $out/p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python tools/pmc_sq_table.py $out
