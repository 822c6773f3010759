set -e
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3
