#!/bin/bash
# round 4, GPU call 3: the stream conv kernel -- correctness, A/B against the team kernel, per-launch times
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_spconv_gpu.py tests/test_native_scan_gpu.py -m gpu -x -q > $OUT/r04_c3_pytest.txt 2>&1
echo "pytest rc $?" >> $OUT/r04_c3_pytest.txt
cd /tmp && export TMPDIR=/tmp
for v in "SG_CONV_STREAM=0" "SG_CONV_STREAM=1" "SG_CONV_STREAM_SNAP=4" "SG_CONV_STREAM_SNAP=16" "SG_CONV_STREAM_MIN=2" "SG_CONV_STREAM_MIN=8" "SG_CONV_STREAM_WPC=2" "SG_CONV_STREAM_WPC=1" "SG_CONV_NBW=1"; do
  echo "== $v: $(env $v timeout 120 python $R/tools/conv_only.py 20 2>&1 | tail -1)" >> $OUT/r04_c3_conv_only.txt
done
for v in "SG_CONV_STREAM=0" "SG_CONV_STREAM=1"; do
  rm -rf /tmp/prof
  env $v timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r -- python $R/tools/conv_only.py 5 > /dev/null 2>&1
  echo "== $v" >> $OUT/r04_c3_conv_seq.txt
  python $R/tools/conv_seq.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) 6 >> $OUT/r04_c3_conv_seq.txt 2>&1
done
env SG_CONV_STREAM=1 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $OUT/r04_c3_bench.json 2> $OUT/r04_c3_bench.err
echo done
