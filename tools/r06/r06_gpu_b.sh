#!/bin/bash
# round-6 GPU call B: conv chain -- tests, A/B latency, per-scan kernel table
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_conv_chain_gpu.py -m gpu -x -q 2>&1 | tail -15 > $OUT/r06b_tests.txt
timeout 900 python -m pytest tests/test_spconv_gpu.py tests/test_scan_contexts_gpu.py tests/test_model_gpu.py tests/test_scan_forward_gpu.py -m gpu -x -q 2>&1 | tail -8 >> $OUT/r06b_tests.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT/r06b_chain_ab.txt
for T in 1 0 1 0; do
  echo "== SG_CONV_CHAIN=$T" >> $OUT/r06b_chain_ab.txt
  SG_CONV_CHAIN=$T timeout 300 python $R/tools/scan_only.py 30 150000 scannet 2>/dev/null | tail -1 >> $OUT/r06b_chain_ab.txt
  SG_CONV_CHAIN=$T timeout 300 python $R/tools/conv_exec_layers.py 150000 10 2>&1 | tail -40 >> $OUT/r06b_chain_ab.txt
done
rm -rf /tmp/prof_scan
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_scan -o r -- python $R/tools/scan_only.py 16 150000 scannet > /dev/null 2>&1
python $R/tools/scan_sequence.py /tmp/prof_scan $OUT/r06b_scan_scannet pointwise_heads_kernel
timeout 600 python $R/bench.py --no-cpu-baseline --no-legs > $OUT/r06b_bench.json 2> $OUT/r06b_bench.err
timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $OUT/r06b_bench20.json 2>> $OUT/r06b_bench.err
echo done
