cd $GRAFT_REPO_ROOT
bash naima_amd/csrc/build.sh $VARIANT > /dev/null 2>&1 || echo "build failed"
timeout 300 python scripts/hs_stamps.py ${W:-cfg3 512} 0.005 40 1 1 2>&1 | grep "^  wave\|^block 0 start" | cut -c1-200
bash naima_amd/csrc/build.sh > /dev/null 2>&1
