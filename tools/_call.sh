cd $GRAFT_REPO_ROOT
python -m pytest tests/test_spconv_gpu.py -x -q 2>&1 | tail -2
python tools/conv_exec_layers.py 150000 10 > gpurun_out/c13_layers.txt 2>&1; tail -1 gpurun_out/c13_layers.txt | cut -c1-150
python tools/scan_only.py 30 2>&1 | tail -1
