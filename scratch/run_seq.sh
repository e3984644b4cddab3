set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for per in 4 2 1; do
  echo "== slices/wg $per"
  GMC_LDS_SLICES_PER_WG=$per python bench.py --mode sequential --graphs-per-gpu 20 --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/seq_$per.err | tee gpurun_out/seq_$per.json
done
mkdir -p gpurun_out/pmc_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/pmc_lds -o lds --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_lds/bench.log 2>&1
ls gpurun_out/pmc_lds
