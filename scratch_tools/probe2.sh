cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "deep_tail" -s 2>&1 | grep -v "^$" | tail -40
