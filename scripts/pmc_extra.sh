#!/bin/bash
# extra PMC passes (instruction cache, wait breakdown): bash scripts/pmc_extra.sh TAG
set -e
tag=${1:-extra}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $out/p1 -- python3 $B > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/p2 -- python3 $B > $out/p2.log 2>&1
rocprofv3 --pmc SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p3 -- python3 $B > $out/p3.log 2>&1
python3 scripts/pmc_summary.py $out/p1 $out/p2 $out/p3 > $out/summary.json
find $out -name '*.csv' -size +2M -delete
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d.items(): print(k,{c:round(x['mean_per_launch']) for c,x in v.items()})
PY
