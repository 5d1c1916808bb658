cd $GRAFT_REPO_ROOT
timeout 300 python scripts/hs_stamps.py cfg3 512 2>&1 | tail -20
bash naima_amd/csrc/build.sh -DHS_SKIP_TAB > /dev/null 2>&1
timeout 300 python scripts/hs_stamps.py cfg3 512 2>&1 | tail -18
bash naima_amd/csrc/build.sh > /dev/null 2>&1
