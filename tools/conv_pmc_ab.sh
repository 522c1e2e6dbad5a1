#!/bin/bash
# HBM-side traffic and L2 hit rate of the conv kernel under the default tile plan and under round 5's
# spatially local plan (SG_UNET_MORTON=1 SG_PLAN_ORDER=1), rocprofv3 --pmc, one counter set per pass,
# kernel-trace only, over tools/conv_only.py 2 (3 backbone forwards of the bench scene).
#   bash tools/conv_pmc_ab.sh <tag>   -> gpurun_out/<tag>_conv_pmc.{txt,json} (default plan),
#                                        gpurun_out/<tag>_conv_pmc_local.{txt,json} (local plan)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {   # name, counters, env...
  local name=$1 set="$2"; shift 2
  rm -rf /tmp/pmc
  env "$@" timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python $R/tools/conv_only.py 2 > /tmp/pmc.log 2>&1
  echo "== $set ($* ; rc $?)" >> $OUT/${name}.txt
  python $R/tools/pmc_summary.py /tmp/pmc gather_conv_persistent_kernel --json $OUT/${name}.json --scans 3 >> $OUT/${name}.txt 2>&1
}
rm -f $OUT/${TAG}_conv_pmc.txt $OUT/${TAG}_conv_pmc.json $OUT/${TAG}_conv_pmc_local.txt $OUT/${TAG}_conv_pmc_local.json
run ${TAG}_conv_pmc FETCH_SIZE SG_PLAN_ORDER=0
run ${TAG}_conv_pmc_local FETCH_SIZE SG_UNET_MORTON=1 SG_PLAN_ORDER=1
run ${TAG}_conv_pmc "TCC_HIT_sum TCC_MISS_sum" SG_PLAN_ORDER=0
run ${TAG}_conv_pmc_local "TCC_HIT_sum TCC_MISS_sum" SG_UNET_MORTON=1 SG_PLAN_ORDER=1
run ${TAG}_conv_pmc WRITE_SIZE SG_PLAN_ORDER=0
echo done
