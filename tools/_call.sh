cd $GRAFT_REPO_ROOT
python -m pytest tests/test_scan_contexts_gpu.py -x -q 2>&1 | grep -E "passed|failed"
python bench.py > gpurun_out/r05_bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_shape.json 2>/dev/null
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-roofline 2>/dev/null > gpurun_out/r05_bench_driver_shape_$i.json; done
