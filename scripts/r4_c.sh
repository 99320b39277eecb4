cd $GRAFT_REPO_ROOT
bash scripts/ablate_classes.sh r4c
