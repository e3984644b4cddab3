# round-3 evidence, ONE rocprofv3 pass per gpurun call (two passes in one call stalled between them once):
#   bash scratch/run_profile_r03.sh stats|fetch|write [TAG]     then, wherever gpurun_out/ is:  python scratch/summarize_r03.py TAG
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GCN_MAXCUT_BENCH_SERIAL_GEN=1   # no forked graph-generation workers under the profiler (they hang now and then once its library has initialised the GPU)
PASS=${1:-stats}
TAG=${2:-r03a}
SMALL="--steps 5 --warmup 1 --no-cpu-baseline --no-sequential --no-other-workloads --no-dp-profile"
case $PASS in
  stats) mkdir -p gpurun_out/prof_$TAG
         rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o run --output-format csv -- python bench.py > gpurun_out/prof_$TAG/bench_line.json 2> gpurun_out/prof_$TAG/err.txt ;;
  fetch) mkdir -p gpurun_out/pmc_fetch_$TAG
         rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG -o run --output-format csv -- python bench.py $SMALL > gpurun_out/pmc_fetch_$TAG/log.txt 2>&1 ;;
  write) mkdir -p gpurun_out/pmc_write_$TAG
         rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG -o run --output-format csv -- python bench.py $SMALL > gpurun_out/pmc_write_$TAG/log.txt 2>&1 ;;
esac
echo $PASS done
