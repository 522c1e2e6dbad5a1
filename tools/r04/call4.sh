#!/bin/bash
# round 4, GPU call 4: stream kernel after the metadata / reducer fixes
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_spconv_gpu.py -m gpu -q > $OUT/r04_c4_pytest.txt 2>&1
echo "pytest rc $?" >> $OUT/r04_c4_pytest.txt
cd /tmp && export TMPDIR=/tmp
for v in "SG_CONV_STREAM=0" "SG_CONV_STREAM=1" "SG_CONV_STREAM_MIN_PAIRS=300" "SG_CONV_STREAM_MIN_PAIRS=1000" "SG_CONV_STREAM_MIN_PAIRS=3000" "SG_CONV_STREAM_MIN=8" "SG_CONV_STREAM_MIN=16" "SG_CONV_STREAM_WPC=2" "SG_CONV_STREAM_SNAP=16"; do
  echo "== $v: $(env $v timeout 120 python $R/tools/conv_only.py 20 2>&1 | tail -1)" >> $OUT/r04_c4_conv_only.txt
done
for v in "SG_CONV_STREAM=1" "SG_CONV_STREAM_MIN=8"; do
  rm -rf /tmp/prof
  env $v timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r -- python $R/tools/conv_only.py 5 > /dev/null 2>&1
  echo "== $v" >> $OUT/r04_c4_conv_seq.txt
  python $R/tools/conv_seq.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) 6 >> $OUT/r04_c4_conv_seq.txt 2>&1
done
echo done
