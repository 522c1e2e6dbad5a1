#!/bin/bash
# One rocprofv3 --pmc pass (kernel-trace only) with the texture-addresser / L1 counters over a
# short tools/conv_only.py run (backbone only): is the sparse-conv kernel bound by the CU's vector-memory path?
#   bash tools/conv_pmc_ta.sh <tag> [ENV=VALUE ...]   -> gpurun_out/<tag>_conv_ta_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/${TAG}_conv_ta_pmc.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmct $OUT
S="TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
env "$@" rocprofv3 --pmc $S --kernel-trace --output-format csv -d /tmp/pmct -- python $R/tools/conv_only.py 2 > /tmp/pmct.log 2>&1
echo "== $S" >> $OUT
python $R/tools/pmc_summary.py /tmp/pmct gather_conv_persistent_kernel >> $OUT 2>&1
