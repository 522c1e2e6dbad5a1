#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_ref_gpu_kernels.py tests/test_native_scan_gpu.py -x -q -m gpu > $OUT/r04_c45_full.txt 2>&1
grep -E "passed|failed" $OUT/r04_c45_full.txt | tail -2 > $OUT/r04_c45_tests.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/tools/host_profile.py 10 kitti > /dev/null 2> /tmp/prof.err
python $R/tools/kernel_stats.py $(find /tmp/prof -name "*kernel_stats.csv" | head -1) 12 40 2>&1 | grep -E "GPU busy|bfs_union" > $OUT/r04_c45_kitti.txt
python $R/tools/host_profile.py 10 kitti 2>/dev/null | head -1 >> $OUT/r04_c45_kitti.txt
echo done
