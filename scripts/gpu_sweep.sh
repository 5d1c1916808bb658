cd $GRAFT_REPO_ROOT
for cfg in "1 2 16" "2 2 8" "2 2 16" "2 1 8" "2 4 8" "3 2 8" "4 2 4"; do set -- $cfg; echo "nsplit=$1 W=$2 C=$3 $(KB_NSPLIT=$1 NH_INT_W=$2 NH_INT_C=$3 python scripts/kbench.py 256 2>&1 | grep 'integrate(IC')"; done
