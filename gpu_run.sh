cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python tools/profile_sweep.py 1024 50; done
python tools/profile_sweep.py 4096 20
rocm-smi --showclocks 2>/dev/null | head -20
