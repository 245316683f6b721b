#!/bin/bash
# ONE script for the evidence the bench line quotes: the rocprofv3 kernel trace (+ stats) and the PMC passes of the same
# command on the same sources.  Runs on the GPU box from the repo root; everything lands under gpurun_out/<name>/:
#   trace/          rocprofv3 --kernel-trace --stats of   bench.py --steps 10 --warmup 2 --no-cpu --no-extras
#   trace_bench_line.json   the JSON line that run printed
#   pass1..7/       one rocprofv3 --pmc run each (counters are never mixed with tracing, one pass per run)
#   csrc_sha16.txt  hash of raytracer_amd/csrc the library was built from
# Afterwards, on the build box:  python scripts/pmc_to_json.py gpurun_out/<name> profiles/r03/<label>
#   -> profiles/pmc_traffic.json (counters + the trace's average launch times, keyed by the hash) and the kept copies.
# usage: scripts/profile.sh <name>
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$R'); from bench import csrc_sha16; print(csrc_sha16())" > $OUT/csrc_sha16.txt
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-extras > $OUT/trace_bench_line.json 2> $OUT/trace.log || { echo "kernel trace failed"; tail -3 $OUT/trace.log; exit 1; }
echo "trace ok"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "FETCH_SIZE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "GRBM_GUI_ACTIVE GRBM_COUNT" \
         "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pass$i.log; exit 1; }
  echo "pass $i ok"
done
python3 $R/scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
