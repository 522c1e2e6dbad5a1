cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_native_scan_gpu.py -x -q -k "bfs or native" > gpurun_out/c20_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/c20_pytest.log | tail -2
cd /tmp && export TMPDIR=/tmp
for cfg in scannet kitti; do
rm -rf /tmp/prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r -- python $GRAFT_REPO_ROOT/tools/scan_only.py 12 150000 $cfg > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/scan_sequence.py /tmp/prof $GRAFT_REPO_ROOT/gpurun_out/c20_${cfg} pointwise_heads_kernel
grep "bfs_union\|bfs_flatten\|bfs_store_root\|bfs_hook\|bfs_compress\|GPU busy" $GRAFT_REPO_ROOT/gpurun_out/c20_${cfg}_top.txt | cut -c1-125
done
cd $GRAFT_REPO_ROOT
python tools/scan_only.py 30 2>&1 | tail -1
SG_BFS_ONE_PASS=1 python tools/scan_only.py 30 2>&1 | tail -1
python tools/scan_only.py 20 150000 kitti 2>&1 | tail -1
SG_BFS_ONE_PASS=1 python tools/scan_only.py 20 150000 kitti 2>&1 | tail -1
