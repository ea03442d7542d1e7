cd $GRAFT_REPO_ROOT; timeout 600 python tools/other_configs.py 2>&1 | tail -5
