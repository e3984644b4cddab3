# round-3 evidence: rocprofv3 kernel stats of the DEFAULT bench command (grouped by kernel and grid), HBM traffic from
# separate --pmc FETCH_SIZE / WRITE_SIZE passes (per kernel and grid, so the SpMM batch sweep's launches separate)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03a}
mkdir -p gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o run --output-format csv -- python bench.py > gpurun_out/prof_$TAG/bench_line.json 2> gpurun_out/prof_$TAG/err.txt
echo stats done
SMALL="--steps 5 --warmup 1 --no-cpu-baseline --no-sequential --no-other-workloads --no-dp-profile"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG -o run --output-format csv -- python bench.py $SMALL > gpurun_out/pmc_fetch_$TAG/log.txt 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG -o run --output-format csv -- python bench.py $SMALL > gpurun_out/pmc_write_$TAG/log.txt 2>&1
echo write done
python scratch/summarize_r03.py $TAG
rm -f gpurun_out/prof_$TAG/run_kernel_trace.csv gpurun_out/prof_$TAG/*/run_kernel_trace.csv gpurun_out/pmc_*_$TAG/run_kernel_trace.csv gpurun_out/pmc_*_$TAG/*/run_kernel_trace.csv
du -sh gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
