cd $GRAFT_REPO_ROOT
for a in "cfg5 256" "cfg5 2048" "cfg1 32"; do timeout 300 python scripts/hs_stamps.py $a 2>&1 | tail -20; done
