#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_native_scan_gpu.py tests/test_parity_at_size.py -m gpu -x -q -k "bfs or softgroup_pp or config4" 2>&1 | tail -4 > $OUT/r06o_tests.txt
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06o_ab.txt
: > $F
for T in 0 1 2 0 1 2; do
  echo "== SG_BFS_THIN=$T" >> $F
  SG_BFS_THIN=$T timeout 300 python $R/tools/scan_only.py 30 150000 scannet 2>/dev/null | tail -1 >> $F
done
echo "== stpls3d_pp with the octree stash" >> $F
timeout 300 python $R/tools/scan_only.py 30 150000 stpls3d_pp 2>/dev/null | tail -1 >> $F
timeout 300 python $R/tools/scan_only.py 30 150000 stpls3d_pp 2>/dev/null | tail -1 >> $F
rm -rf /tmp/prof_scan
SG_BFS_THIN=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_scan -o r -- python $R/tools/scan_only.py 16 150000 scannet > /dev/null 2>&1
python $R/tools/scan_sequence.py /tmp/prof_scan $OUT/r06o_scan_thin1 pointwise_heads_kernel
grep "bfs_emit_kernel" $OUT/r06o_scan_thin1_sequence.txt | head -3 >> $F
rm -rf /tmp/prof_scan
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_scan -o r -- python $R/tools/scan_only.py 16 150000 stpls3d_pp > /dev/null 2>&1
python $R/tools/scan_sequence.py /tmp/prof_scan $OUT/r06o_scan_stpls3d_pp pointwise_heads_kernel
grep "octree" $OUT/r06o_scan_stpls3d_pp_top.txt >> $F
echo done
cd $R
timeout 600 python -m pytest tests/test_data_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $OUT/r06o_tests.txt
python - >> $OUT/r06o_ab.txt 2>/dev/null <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from softgroup_amd import synthetic
from softgroup_amd.data import collate_device, make_item, prefetch_device
xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
sample = make_item(xyz, rgb, 50, None, inst, 'synthetic_0000')
model = synthetic.build_model(seed=0)
with torch.no_grad():
    for _ in range(3):
        model(collate_device([sample])).resolve()
    for b in prefetch_device([[sample]] * 3):
        model(b).resolve()
    for name, it in (('in line', lambda: (collate_device([sample]) for _ in range(20))), ('prefetched', lambda: prefetch_device([[sample]] * 20))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rets = [model(b) for b in it()]
        for r in rets:
            r.resolve()
        torch.cuda.synchronize()
        print('with_h2d', name, round((time.perf_counter() - t0) / 20 * 1e3, 3), 'ms per scan')
PY
echo done2
