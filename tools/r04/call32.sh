#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_ref_gpu_kernels.py tests/test_native_scan_gpu.py tests/test_lazy_results.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4 > $OUT/r04_c32_tests.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --contexts 1 --steps 10 --warmup 8 --no-cpu-baseline --no-legs --no-roofline > $OUT/r04_c32_bench_rocprof.json 2> /tmp/prof.err
python $R/tools/kernel_stats.py $(find /tmp/prof -name "*kernel_stats.csv" | head -1) 44 60 2>&1 | grep -E "GPU busy|bq_|copyBuffer|fill|CatArray" > $OUT/r04_c32_bq.txt
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/tools/train_profile.py 10 plain16 > /dev/null 2> /tmp/prof.err
python $R/tools/kernel_stats.py $(find /tmp/prof -name "*kernel_stats.csv" | head -1) 13 60 > $OUT/r04_c32_train_kernel_top.txt 2>&1
timeout 300 python $R/tools/train_profile.py 20 plain16 > $OUT/r04_c32_train_plain16.txt 2>&1
echo done
