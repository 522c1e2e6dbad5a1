cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -k "pointwise_heads" 2>&1 | grep -E "passed|failed"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r -- python $GRAFT_REPO_ROOT/tools/scan_only.py 12 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/scan_sequence.py /tmp/prof $GRAFT_REPO_ROOT/gpurun_out/c22_scan pointwise_heads_kernel
grep "pointwise_heads\|GPU busy\|scan_block_sums\|scan_apply\|scan_reduce" $GRAFT_REPO_ROOT/gpurun_out/c22_scan_top.txt | cut -c1-125
