set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  mkdir -p gpurun_out/pmc2/s$i
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc2/s$i -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe > gpurun_out/pmc2/s$i/log.txt 2>&1 || echo "set $i failed"
  echo "set $i done"
done
