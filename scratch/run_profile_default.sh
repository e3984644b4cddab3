set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r01g
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01g -o run --output-format csv -- python bench.py > gpurun_out/prof_r01g/bench_line.json 2> gpurun_out/prof_r01g/err.txt
tail -c 900 gpurun_out/prof_r01g/bench_line.json
rm -f gpurun_out/prof_r01g/run_kernel_trace.csv
ls -la gpurun_out/prof_r01g
