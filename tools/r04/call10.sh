#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_variants_gpu.py tests/test_native_scan_gpu.py tests/test_parity_at_size.py -m gpu -q > $OUT/r04_c10_pytest.txt 2>&1
echo "pytest rc $?" >> $OUT/r04_c10_pytest.txt
cd /tmp && export TMPDIR=/tmp
for cfg in stpls3d_pp kitti; do
  rm -rf /tmp/prof_$cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o r -- python $R/tools/host_profile.py 10 $cfg > $OUT/r04_c10_${cfg}_host_profile.txt 2>&1
  cp $(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1) $OUT/r04_c10_${cfg}_kernel_stats.csv
  python $R/tools/kernel_stats.py $OUT/r04_c10_${cfg}_kernel_stats.csv 12 40 > $OUT/r04_c10_${cfg}_kernel_top.txt 2>&1
done
python $R/tools/dense_profile.py > $OUT/r04_c10_dense.txt 2>&1
echo done
