set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/tl && mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o run --output-format csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sequential --no-probe > gpurun_out/tl/line.json 2> gpurun_out/tl/err.txt
python scratch/timeline.py gpurun_out/tl
head -3 gpurun_out/tl/*/*memory_copy_trace.csv || true
rm -f gpurun_out/tl/*/*kernel_trace.csv gpurun_out/tl/*/*memory_copy_trace.csv
