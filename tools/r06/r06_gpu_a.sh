#!/bin/bash
# round-6 GPU call A: BFS thin levels (tests + A/B), pool test, 10 default bench runs + 5 driver-shaped runs
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests/test_ops_gpu.py tests/test_scan_contexts_gpu.py tests/test_native_scan_gpu.py -m gpu -x -q 2>&1 | tail -5 > $OUT/r06a_tests.txt
cd /tmp && export TMPDIR=/tmp
for T in 1 0; do
  echo "== SG_BFS_THIN=$T" >> $OUT/r06a_thin_ab.txt
  for CFG in scannet kitti; do
    SG_BFS_THIN=$T python $R/tools/scan_only.py 30 150000 $CFG 2>/dev/null | tail -1 >> $OUT/r06a_thin_ab.txt
  done
done
rm -rf /tmp/prof_scan
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_scan -o r -- python $R/tools/scan_only.py 16 150000 scannet > /dev/null 2>&1
python $R/tools/scan_sequence.py /tmp/prof_scan $OUT/r06a_scan_scannet pointwise_heads_kernel
: > $OUT/r06a_inflight_modes.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  python $R/bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default run', '$i', 'ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency_ms', d.get('latency_ms'), 'identical', d.get('timed_results_identical'))" >> $OUT/r06a_inflight_modes.txt
done
for i in 1 2 3 4 5; do
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver shape (--steps 20 --warmup 5) run', '$i', 'ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency_ms', d.get('latency_ms'))" >> $OUT/r06a_inflight_modes.txt
done
echo done
