cd $GRAFT_REPO_ROOT
echo "A default";            python tools/scan_only.py 30 2>&1 | tail -1
echo "B SG_EARLY_COPY=0";    SG_EARLY_COPY=0 python tools/scan_only.py 30 2>&1 | tail -1
echo "C HSA_ENABLE_SDMA=1";  HSA_ENABLE_SDMA=1 python tools/scan_only.py 30 2>&1 | tail -1
echo "D HSA_ENABLE_SDMA=0";  HSA_ENABLE_SDMA=0 python tools/scan_only.py 30 2>&1 | tail -1
echo "E SDMA=1 EARLY=0";     HSA_ENABLE_SDMA=1 SG_EARLY_COPY=0 python tools/scan_only.py 30 2>&1 | tail -1
echo "A default";            python tools/scan_only.py 30 2>&1 | tail -1
