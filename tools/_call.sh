cd $GRAFT_REPO_ROOT
run() { env "$@" python tools/conv_exec_layers.py 150000 10 > gpurun_out/c4_tmp.txt 2>&1; echo "$* :: $(tail -1 gpurun_out/c4_tmp.txt | cut -c1-175)"; }
run SG_UNET_MORTON=0 SG_PLAN_ORDER=0
run SG_UNET_MORTON=1 SG_PLAN_ORDER=1 SG_PLAN_SB=4096
run SG_UNET_MORTON=1 SG_PLAN_ORDER=1 SG_PLAN_SB=8192
run SG_UNET_MORTON=1 SG_PLAN_ORDER=1 SG_PLAN_SB=16384
run SG_UNET_MORTON=1 SG_PLAN_ORDER=1
cp gpurun_out/c4_tmp.txt gpurun_out/c4_layers_auto.txt
run SG_UNET_MORTON=1 SG_PLAN_ORDER=2
run SG_UNET_MORTON=1 SG_PLAN_ORDER=2 SG_CONV_STATIC=0
run SG_UNET_MORTON=0 SG_PLAN_ORDER=1
run SG_UNET_MORTON=0 SG_PLAN_ORDER=0
cp gpurun_out/c4_tmp.txt gpurun_out/c4_layers_legacy.txt
