set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gcn-max-cut_amd/lib
GCN_MAXCUT_LIB=$L/libgcnmaxcut_${TESTLIB:-t512}.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_goldens.py -m gpu -q --timeout 900 -k "not bench" 2>&1 | tail -5
bash scratch/run_var.sh
