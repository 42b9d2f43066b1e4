cd $GRAFT_REPO_ROOT
AGG_VARIANTS=7,12,9,13,6 python tools/agg_time.py rounds
