cd $GRAFT_REPO_ROOT
python -m pytest tests/test_native_scan_gpu.py tests/test_dropin_gpu.py tests/test_variants_gpu.py tests/test_ops_gpu.py -x -q > gpurun_out/c16_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/c16_pytest.log | tail -3
