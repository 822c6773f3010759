set -e
cd $GRAFT_REPO_ROOT
python3 tests/diag/batch_queue_phase.py
python3 tests/diag/c2_phases.py | tail -1
bash tests/diag/r4_lanes7.sh
