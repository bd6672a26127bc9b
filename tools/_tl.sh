timeout 2400 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
