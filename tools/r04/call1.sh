#!/bin/bash
# round 4, GPU call 1: GPU suite on the new defaults, conv A/B, per-layer table, config 4 / 5 profiles
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r04_c1_pytest.txt 2>&1
echo "pytest rc $?" >> $OUT/r04_c1_pytest.txt
cd /tmp && export TMPDIR=/tmp
ab() {
  env $1 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'conv ms/scan', r['kernel_ms_per_scan'], 'launches', r['launches_per_scan'], 'bound', r['bound'], 'frac', r['frac'], 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'], 'one-at-a-time', d['ms_per_step_one_scan_at_a_time'], 'latency', d['latency_ms'], d['stages_ms'])"
}
for v in "SG_X=0" "SG_CONV_TARGET=2048" "SG_CONV_TARGET=4096" "SG_CONV_SPLIT_MIN_CIN=32" "SG_CONV_COMBINE=0" "SG_X=1"; do
  ab "$v" >> $OUT/r04_c1_conv_ab.txt 2>&1
done
python $R/tools/conv_layers.py > $OUT/r04_c1_conv_layers.txt 2>&1
for cfg in stpls3d_pp kitti; do
  rm -rf /tmp/prof_$cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o r -- python $R/tools/host_profile.py 10 $cfg > $OUT/r04_${cfg}_host_profile.txt 2>&1
  cp $(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1) $OUT/r04_${cfg}_kernel_stats.csv
  python $R/tools/kernel_stats.py $OUT/r04_${cfg}_kernel_stats.csv 12 50 > $OUT/r04_${cfg}_kernel_top.txt 2>&1
done
echo done
