cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_gpu.py -x -q -s -k "gradients_match_reference" > gpurun_out/c18_pytest.log 2>&1
grep -E "passed|failed|error|tensors|Error|assert" gpurun_out/c18_pytest.log | tail -12
