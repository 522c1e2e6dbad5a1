cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_spconv_gpu.py tests/test_native_scan_gpu.py tests/test_scan_contexts_gpu.py -x -q > gpurun_out/c21_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/c21_pytest.log | tail -2
python tools/scan_only.py 30 2>&1 | tail -1
python tools/scan_only.py 20 150000 kitti 2>&1 | tail -1
python tools/scan_only.py 12 150000 stpls3d_pp 2>&1 | tail -1
