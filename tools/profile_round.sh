#!/bin/bash
# The rocprofv3 passes behind profiles/r03_*: run on the GPU box from the repository root (gpurun).  Counter passes are separate
# from the kernel trace (MI355X_MICROARCH.md); every pass profiles `python3 bench.py ...` directly.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=20     # what bench.py sets for itself; exported here because the profiler starts the runtime first
B="python3 $R/bench.py --no-cpu-baseline --no-verify"
F1="--gop 0 --frames-in-flight 1"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o run -- $B --steps 10 --warmup 2 --no-upload > $O/trace_default.log 2>&1
echo trace_default done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f1 -o run -- $B --steps 20 --warmup 3 $F1 > $O/trace_f1.log 2>&1
echo trace_f1 done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $B --steps 3 --warmup 1 $F1 > $O/pmc_fetch.log 2>&1
echo pmc_fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $B --steps 3 --warmup 1 $F1 > $O/pmc_write.log 2>&1
echo pmc_write done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -o run -- $B --steps 2 --warmup 1 $F1 > $O/pmc_sq.log 2>&1
echo pmc_sq done
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq2 -o run -- $B --steps 2 --warmup 1 $F1 > $O/pmc_sq2.log 2>&1 || echo "pmc_sq2 failed (counters not all available)"
echo pmc_sq2 done
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 4 > $O/pmc_traffic.json
python3 $R/tools/pmc_summary.py $O/r03_pmc $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2
# keep what is merged back small: the per-dispatch traces of the kernel-trace passes and the raw counter dumps are not needed
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +2M -delete
ls -R $O | head -60
