# HBM traffic counters of the bench kernels (separate --pmc passes) + in-kernel phase stamps / wall-clock marks
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02f}
mkdir -p gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG -o run --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sequential > gpurun_out/pmc_fetch_$TAG/log.txt 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG -o run --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sequential > gpurun_out/pmc_write_$TAG/log.txt 2>&1
echo write done
rm -f gpurun_out/pmc_*_$TAG/run_kernel_trace.csv gpurun_out/pmc_*_$TAG/*/run_kernel_trace.csv
python scratch/stamps.py > gpurun_out/stamps_$TAG.txt 2>&1 || true
tail -9 gpurun_out/stamps_$TAG.txt
