# round-2 profile: kernel stats of the DEFAULT bench command, HBM traffic counters, in-kernel phase stamps
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02a}
mkdir -p gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o run --output-format csv -- python bench.py > gpurun_out/prof_$TAG/bench_line.json 2> gpurun_out/prof_$TAG/err.txt
tail -c 1500 gpurun_out/prof_$TAG/bench_line.json
rm -f gpurun_out/prof_$TAG/run_kernel_trace.csv gpurun_out/prof_$TAG/*/run_kernel_trace.csv
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG -o run --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sequential > gpurun_out/pmc_fetch_$TAG/log.txt 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG -o run --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sequential > gpurun_out/pmc_write_$TAG/log.txt 2>&1
echo write done
rm -f gpurun_out/pmc_*_$TAG/run_kernel_trace.csv gpurun_out/pmc_*_$TAG/*/run_kernel_trace.csv
python scratch/stamps.py > gpurun_out/stamps_$TAG.txt 2>&1 || true
cat gpurun_out/stamps_$TAG.txt | tail -4
du -sh gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
