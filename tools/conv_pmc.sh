#!/bin/bash
# Where does the sparse-conv kernel wait?  rocprofv3 --pmc passes (kernel-trace only) over
# tools/conv_layers.py, aggregated per kernel variant and grid size by tools/pmc_summary.py.
#   bash tools/conv_pmc.sh <tag> [ENV=VALUE ...]      -> gpurun_out/<tag>_conv_wait_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/${TAG}_conv_wait_pmc.txt
cd /tmp && export TMPDIR=/tmp
rm -f $OUT
for S in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
         "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum TCC_REQ_sum"; do
  rm -rf /tmp/pmcw
  env "$@" rocprofv3 --pmc $S --kernel-trace --output-format csv -d /tmp/pmcw -- python $R/tools/conv_layers.py 150000 > /tmp/pmcw.log 2>&1
  echo "== $S" >> $OUT
  python $R/tools/pmc_summary.py /tmp/pmcw gather_conv_persistent_kernel >> $OUT 2>&1
done
