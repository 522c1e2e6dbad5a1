#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_conv_chain_gpu.py -m gpu -x -q 2>&1 | tail -5 > $OUT/r06c_tests.txt
cd /tmp && export TMPDIR=/tmp
rm -f /tmp/chain.bin
SG_CHAIN_TRACE=/tmp/chain.bin timeout 300 python $R/tools/conv_only.py 2 > /dev/null 2>&1
python $R/tools/chain_trace.py /tmp/chain.bin > $OUT/r06c_chain_trace.txt 2>&1
echo done
