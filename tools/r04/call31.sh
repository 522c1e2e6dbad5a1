#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_spconv_gpu.py tests/test_native_scan_gpu.py tests/test_data_gpu.py -x -q -m gpu 2>&1 | tail -4 > $OUT/r04_c31_tests.txt
timeout 300 python tools/copy_sites.py all > $OUT/r04_c31_torch_ops.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --contexts 1 --steps 10 --warmup 8 --no-cpu-baseline --no-legs --no-roofline > $OUT/r04_c31_bench_rocprof.json 2> /tmp/prof.err
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $OUT/r04_c31_kernel_stats.csv
echo done
