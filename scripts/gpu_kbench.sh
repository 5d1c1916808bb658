cd $GRAFT_REPO_ROOT
python scripts/kbench.py 256 2>&1 | tail -8
for c in 8 16; do echo -n "C=$c "; NH_INT_C=$c python scripts/kbench.py 256 2>&1 | grep -E "integrate\(IC"; done
