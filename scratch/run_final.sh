# end-of-round evidence: full GPU suite, smoke(), the default bench line, its rocprofv3 stats by grid
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/tests_final.log 2>&1 || true
tail -3 gpurun_out/tests_final.log
python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 300 gpurun_out/bench_final.json; echo
bash scratch/run_profile_r02b.sh r02c > gpurun_out/profile_r02c.log 2>&1
head -12 gpurun_out/prof_r02c/kernel_stats_by_grid.csv
