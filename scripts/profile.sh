#!/bin/bash
# Profile passes for profiles/rN (run on the GPU box through gpurun, from the repo root):
#   gpurun --timeout 900 -- 'bash scripts/profile.sh v5'
# Pass 1 kernel trace + stats; passes 2-4 PMC counters, each in a run of its own (no trace domains mixed in).
set -e
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --steps 4 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $B > $out/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
    --output-format csv -d $out/pmc_sq -- python3 $B > $out/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $B > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $B > $out/pmc_write.log 2>&1
python3 scripts/pmc_summary.py $out/pmc_sq $out/pmc_fetch $out/pmc_write > $out/pmc_summary.json
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
find $out -name '*.csv' -size +2M -delete
cat $out/kernel_stats.csv
