#!/bin/bash
# Collects the round's measurement set on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02
# -> gpurun_out/<tag>_bench.json, _bench_under_rocprof.json, _kernel_stats.csv, _conv_by_grid.txt,
#    _conv_pmc.{txt,json} (separate --pmc passes, --kernel-trace only), _train_* (training kernels),
#    with CALIB=1 also _calib.txt (FETCH/WRITE_SIZE on known byte counts).  Copy what should be
#    judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# kernel trace + stats of the same command, one scan at a time (--contexts 1: kernel durations are
# not stretched by overlapping scans, so the conv kernel's average agrees with roofline.avg_launch_us);
# forwards in the trace: 8 warm-up + 10 timed + 10 + 10 + 5 stage + 1 + 5 roofline = 49
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --contexts 1 --steps 10 --warmup 8 --no-cpu-baseline --no-legs > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/prof.err
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python $R/tools/kernel_stats.py $OUT/${TAG}_kernel_stats.csv 49 60 > $OUT/${TAG}_kernel_top.txt 2>&1
python $R/tools/conv_by_grid.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) 49 > $OUT/${TAG}_conv_by_grid.txt 2>&1
# the same with 4 scans in flight (the timed region's mode): kernels of different scans overlap
rm -rf /tmp/prof4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o r -- python $R/bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-legs --no-roofline > $OUT/${TAG}_bench_under_rocprof_4ctx.json 2> /tmp/prof4.err
cp $(find /tmp/prof4 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_4ctx.csv
rm -f $OUT/${TAG}_conv_pmc.txt $OUT/${TAG}_conv_pmc.json
for S in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $S --kernel-trace --output-format csv -d /tmp/pmc -- python $R/tools/conv_layers.py 150000 > /tmp/pmc.log 2>&1
  echo "== $S" >> $OUT/${TAG}_conv_pmc.txt
  python $R/tools/pmc_summary.py /tmp/pmc gather_conv_persistent_kernel --json $OUT/${TAG}_conv_pmc.json --scans 8 >> $OUT/${TAG}_conv_pmc.txt 2>&1
done
# training-side kernels (bf16 forward / deterministic wgrad): per-level table + kernel stats
python $R/tools/train_conv_bench.py > $OUT/${TAG}_train_conv.txt 2>/dev/null < /dev/null
rm -rf /tmp/proft
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -o r -- python $R/tools/train_conv_bench.py > /dev/null 2> /tmp/proft.err < /dev/null
F=$(find /tmp/proft -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp "$F" $OUT/${TAG}_train_kernel_stats.csv; fi
python $R/tools/train_step_bench.py 2>/dev/null < /dev/null | grep "ms/step" > $OUT/${TAG}_train_step.txt
if [ -n "$CALIB" ]; then
  rm -f $OUT/${TAG}_calib.txt
  for S in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cal
    rocprofv3 --pmc $S --kernel-trace --output-format csv -d /tmp/cal -- $R/tools/micro/fetch_calib >> $OUT/${TAG}_calib.txt 2>/dev/null
    echo "== $S" >> $OUT/${TAG}_calib.txt
    python $R/tools/pmc_summary.py /tmp/cal _kernel >> $OUT/${TAG}_calib.txt 2>&1
  done
fi
echo done
