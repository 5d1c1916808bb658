cd $GRAFT_REPO_ROOT
for c in 16 8; do for tw in 64 32 22 16; do echo "C=$c TW=$tw $(NH_SYN_C=$c NH_SYN_TW=$tw python scripts/kbench.py 256 2>&1 | grep synchrotron)"; done; done
