cd $GRAFT_REPO_ROOT
for w in 8 16 32; do
echo "SG_BFS_BIG_WGS=$w kitti: $(SG_BFS_BIG_WGS=$w python tools/scan_only.py 16 150000 kitti 2>&1 | tail -1)"
echo "SG_BFS_BIG_WGS=$w stpls3d: $(SG_BFS_BIG_WGS=$w python tools/scan_only.py 10 150000 stpls3d_pp 2>&1 | tail -1)"
done
