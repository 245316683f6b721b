#!/bin/bash
# Collect rocprofv3 PMC passes (one run per pass, as the microarch guide prescribes) for the bench workload.
# usage: scripts/pmc.sh <outdir-under-gpurun_out> ; runs on the GPU box from the repo root.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "FETCH_SIZE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pass$i.log; exit 1; }
  echo "pass $i ok"
done
python3 $R/scripts/pmc_summary.py $OUT > $OUT/summary.txt
