cd $GRAFT_REPO_ROOT
for a in "cfg3 512" "cfg5 256" "cfg1 32" "cfg2 256"; do timeout 300 python scripts/hs_stamps.py $a 2>&1 | tail -5; done
