cd $GRAFT_REPO_ROOT
python tools/ab_streams.py
