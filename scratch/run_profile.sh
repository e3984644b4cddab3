# round profile: kernel stats + HBM traffic counters of the default bench command
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01e}
mkdir -p gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq_$TAG
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 3000 gpurun_out/bench_$TAG.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o run --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG/log.txt 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$TAG -o run --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch_$TAG/log.txt 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$TAG -o run --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write_$TAG/log.txt 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_sq_$TAG -o run --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq_$TAG/log.txt 2>&1
echo sq done
find gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG -name "*.csv" | head
