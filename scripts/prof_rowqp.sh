#!/bin/bash
# rocprofv3 kernel trace + PMC passes of kernel R: bash scripts/prof_rowqp.sh TAG N B
set -e
tag=$1; N=$2; B=$3; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python3 scripts/run_rowqp.py $N $B 5 > $out/run.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/run_rowqp.py $N $B 5 > $out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/p1 -- python3 scripts/run_rowqp.py $N $B 3 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $out/p2 -- python3 scripts/run_rowqp.py $N $B 3 > $out/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$out/kt/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'admpc' in r['Name']: print('%-30s calls %s avg %.1f us' % (r['Name'].split('admpc_')[-1][:28], r['Calls'], float(r['AverageNs'])/1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ('p1','p2'):
    for f in glob.glob('$out/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rowqp' in r['Kernel_Name']: acc[r['Counter_Name']][r['Dispatch_Id']].append(float(r['Counter_Value']))
for c, d in sorted(acc.items()):
    vals = [sum(v) for v in d.values()]
    print('%-24s per launch %.4g' % (c, sum(vals)/len(vals)))
PY
find $out -name '*.csv' -size +2M -delete
cat $out/run.log
