#!/bin/bash
# round 4, GPU call 2: native scan driver tests + A/B, kernel trace of the bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_native_scan_gpu.py tests/test_model_gpu.py tests/test_variants_gpu.py tests/test_parity_at_size.py -m gpu -x -q > $OUT/r04_c2_pytest.txt 2>&1
echo "pytest rc $?" >> $OUT/r04_c2_pytest.txt
cd /tmp && export TMPDIR=/tmp
ab() {
  env $1 python $R/bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'conv ms/scan', r['kernel_ms_per_scan'], 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'], 'one-at-a-time', d['ms_per_step_one_scan_at_a_time'], 'latency', d['latency_ms'], d['stages_ms'])"
}
for v in "SG_NATIVE_SCAN=1" "SG_NATIVE_SCAN=0" "SG_NATIVE_SCAN=1" "SG_NATIVE_SCAN=0"; do
  ab "$v" >> $OUT/r04_c2_ab.txt 2>&1
done
STEPS=160 ab "SG_NATIVE_SCAN=1" >> $OUT/r04_c2_ab.txt 2>&1
STEPS=160 ab "SG_NATIVE_SCAN=0" >> $OUT/r04_c2_ab.txt 2>&1
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --contexts 1 --steps 10 --warmup 8 --no-cpu-baseline --no-legs > $OUT/r04_c2_bench_under_rocprof.json 2> /tmp/prof.err
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $OUT/r04_c2_kernel_stats.csv
python $R/tools/kernel_stats.py $OUT/r04_c2_kernel_stats.csv 49 400 > $OUT/r04_c2_kernel_top.txt 2>&1
python $R/tools/conv_by_grid.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) 49 > $OUT/r04_c2_conv_by_grid.txt 2>&1
echo done
