cd $GRAFT_REPO_ROOT
python scripts/partial_rm_cost.py C3 2>&1 | tail -1
VPFX_NO_ZPROFILE=1 python scripts/partial_rm_cost.py C3 2>&1 | tail -1
python scripts/partial_rm_cost.py C3 0 9 2>&1 | tail -1
VPFX_NO_ZPROFILE=1 python scripts/partial_rm_cost.py C3 0 9 2>&1 | tail -1
