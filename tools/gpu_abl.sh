cd $GRAFT_REPO_ROOT; timeout 600 python tools/pcie_rate.py 2>&1 | tail -4
