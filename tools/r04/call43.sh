#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_native_scan_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_variants_gpu.py tests/test_ref_forward_golden.py -x -q -m gpu > $OUT/r04_c43_full.txt 2>&1
grep -E "passed|failed" $OUT/r04_c43_full.txt | tail -2 > $OUT/r04_c43_tests.txt
cd /tmp && export TMPDIR=/tmp
for w in kitti stpls3d_pp; do
python $R/tools/host_profile.py 10 $w 2>/dev/null | head -1 >> $OUT/r04_c43_legs.txt
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/tools/host_profile.py 10 $w > /dev/null 2> /tmp/prof.err
python $R/tools/kernel_stats.py $(find /tmp/prof -name "*kernel_stats.csv" | head -1) 12 16 > $OUT/r04_c43_${w}_top.txt 2>&1
done
python $R/tools/dense_profile.py > $OUT/r04_c43_dense.txt 2>&1
echo done
