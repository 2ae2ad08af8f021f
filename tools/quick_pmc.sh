#!/bin/bash
# counter passes on a subset of the chain: tools/quick_pmc.sh <tag> <only-stages> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; only=$2; shift 2
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=20
B="python3 $R/bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 --gop 0 --frames-in-flight 1 --only $only $@"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/sq -o run -- $B > $O/sq.log 2>&1 || echo sq failed
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/sq2 -o run -- $B > $O/sq2.log 2>&1 || echo sq2 failed
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $B > $O/fetch.log 2>&1 || echo fetch failed
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- $B > $O/write.log 2>&1 || echo write failed
python3 $R/tools/pmc_summary.py $O/pmc $O/sq $O/sq2 $O/fetch $O/write
find $O -name "*counter_collection.csv" -size +2M -delete
cat $O/pmc_summary.json
