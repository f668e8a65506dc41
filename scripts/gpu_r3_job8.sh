cd $GRAFT_REPO_ROOT
export VPFX_RM_FLAT=1
STEPS=20 bash scripts/gpu_ab.sh 2>&1 | head -4
