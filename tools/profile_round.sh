#!/bin/bash
# Collects the round's measurement set on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r03
# -> gpurun_out/<tag>_bench.json, _bench_under_rocprof.json, _kernel_stats.csv, _kernel_top.txt,
#    _conv_by_grid.txt, _scan_<config>_{sequence,top,latency}.txt, _conv_exec_layers.txt, _conv_layers.txt,
#    _train_conv.txt, _train_step.txt and, with PMC=1,
#    _conv_pmc.{txt,json} (FETCH_SIZE / WRITE_SIZE of the conv kernel over tools/conv_only.py, one
#    --pmc pass each, kernel-trace only).  Copy what should be judged into profiles/.
# PMC passes are slow on this pool (minutes each, every kernel is serialised): they are opt-in and
# run over the backbone-only driver; the TA/TCP counter sets did not finish within 5 minutes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# kernel trace + stats of the same command, one scan at a time (--contexts 1: kernel durations are
# not stretched by overlapping scans, so the conv kernel's average agrees with roofline.avg_launch_us);
# forwards in the trace: counted by the tools from a once-per-scan kernel (bfs_union_kernel)
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --contexts 1 --steps 10 --warmup 8 --no-cpu-baseline --no-legs > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/prof.err
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python $R/tools/kernel_stats.py $OUT/${TAG}_kernel_stats.csv auto 60 > $OUT/${TAG}_kernel_top.txt 2>&1
python $R/tools/conv_by_grid.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) auto > $OUT/${TAG}_conv_by_grid.txt 2>&1
# one scan at a time and NOTHING else in the process (the bench run above also times its legs on the operator
# path): launch sequence of one scan and the per-kernel table per scan, per configuration
for CFG in scannet stpls3d_pp kitti; do
  rm -rf /tmp/prof_scan
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_scan -o r -- python $R/tools/scan_only.py 16 150000 $CFG > /dev/null 2>&1
  python $R/tools/scan_sequence.py /tmp/prof_scan $OUT/${TAG}_scan_${CFG} pointwise_heads_kernel
  python $R/tools/scan_only.py 30 150000 $CFG 2>/dev/null | tail -1 > $OUT/${TAG}_scan_${CFG}_latency.txt
done
python $R/tools/conv_exec_layers.py 150000 10 > $OUT/${TAG}_conv_exec_layers.txt 2>&1
if [ -z "$QUICK" ]; then      # QUICK=1: bench line + kernel trace only
python $R/tools/conv_layers.py > $OUT/${TAG}_conv_layers.txt 2>&1
python $R/tools/train_conv_bench.py > $OUT/${TAG}_train_conv.txt 2>/dev/null < /dev/null
python $R/tools/train_step_bench.py 2>/dev/null < /dev/null | grep "ms/step" > $OUT/${TAG}_train_step.txt
fi
if [ -n "$PMC" ]; then
  rm -f $OUT/${TAG}_conv_pmc.txt $OUT/${TAG}_conv_pmc.json
  for S in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pmc
    timeout 300 rocprofv3 --pmc $S --kernel-trace --output-format csv -d /tmp/pmc -- python $R/tools/conv_only.py 2 > /tmp/pmc.log 2>&1
    echo "== $S (rc $?)" >> $OUT/${TAG}_conv_pmc.txt
    python $R/tools/pmc_summary.py /tmp/pmc gather_conv_persistent_kernel --json $OUT/${TAG}_conv_pmc.json --scans 3 >> $OUT/${TAG}_conv_pmc.txt 2>&1
  done
fi
echo done
