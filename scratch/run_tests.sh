set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 ${PYTEST_ARGS:-} > gpurun_out/tests.log 2>&1 || true
tail -60 gpurun_out/tests.log
