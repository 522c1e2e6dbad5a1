#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06n_d2h.txt
: > $F
for W in 0 8 16 32 64; do
  echo "== SG_SCAN_D2H_WGS=$W" >> $F
  SG_SCAN_D2H_WGS=$W timeout 300 python $R/tools/scan_only.py 30 150000 scannet 2>/dev/null | tail -1 >> $F
  rm -rf /tmp/prof_scan
  SG_SCAN_D2H_WGS=$W timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_scan -o r -- python $R/tools/scan_only.py 16 150000 scannet > /dev/null 2>&1
  python $R/tools/scan_sequence.py /tmp/prof_scan $OUT/r06n_scan_w$W pointwise_heads_kernel
  grep "d2h_copy\|copyBuffer.*dur  *[0-9][0-9][0-9]\|bfs_seed\|bfs_edge_rec\|bfs_emit_kernel" $OUT/r06n_scan_w${W}_sequence.txt | head -8 >> $F
  head -1 $OUT/r06n_scan_w${W}_top.txt >> $F
done
for W in 0 16 0 16; do
  echo "== bench SG_SCAN_D2H_WGS=$W" >> $F
  SG_SCAN_D2H_WGS=$W python $R/bench.py --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency_ms', d.get('latency_ms'))" >> $F
done
echo done
