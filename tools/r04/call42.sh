#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_native_scan_gpu.py tests/test_ops_gpu.py tests/test_ref_gpu_kernels.py tests/test_variants_gpu.py -x -q -m gpu > $OUT/r04_c42_full.txt 2>&1
grep -E "passed|failed" $OUT/r04_c42_full.txt | tail -2 > $OUT/r04_c42_tests.txt
cd /tmp && export TMPDIR=/tmp
for w in kitti stpls3d_pp; do
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/tools/host_profile.py 10 $w > $OUT/r04_c42_${w}_host.txt 2> /tmp/prof.err
python $R/tools/kernel_stats.py $(find /tmp/prof -name "*kernel_stats.csv" | head -1) 12 14 > $OUT/r04_c42_${w}_top.txt 2>&1
done
echo done
