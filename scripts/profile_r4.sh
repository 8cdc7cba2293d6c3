#!/bin/bash
# Round-4 profile of one bench workload: bash scripts/profile_r4.sh TAG N B DTYPE [extra bench.py args]
#   bench line (20 steps), then rocprofv3 passes of the SAME command (20 + 5 steps: shorter runs do not reach the clocks of the bench line -- 3 % at N = 40), every pass with --no-two-in-flight so that the trace holds
#   exactly the timed single-stream steps: kernel trace + stats, two SQ counter passes, FETCH_SIZE and WRITE_SIZE in passes of their own
#   (MI355X_MICROARCH.md: TCC slots).  Writes gpurun_out/TAG/{bench.json, kernel_stats.csv, pmc_summary.json}; copy what is to be
#   judged into profiles/r4/ as TAG_bench.json, TAG_kernel_stats.csv, TAG_pmc_summary.json (tests/test_profiles.py checks that the three agree).
set -e
tag=$1; N=$2; B=$3; DT=$4; shift 4
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
ARGS="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-in-flight --no-tight-stop --no-live-traffic --horizon $N --batch-per-gpu $B --dtype $DT $*"
python3 bench.py --steps 20 --warmup 5 --horizon $N --batch-per-gpu $B --dtype $DT $* > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $ARGS > $out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/p1 -- python3 $ARGS > $out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p2 -- python3 $ARGS > $out/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -- python3 $ARGS > $out/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -- python3 $ARGS > $out/pw.log 2>&1
cp $(ls $out/kt/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
case " $* " in *" --gp "*) VAR=gp;; *) VAR="";; esac
python3 scripts/pmc_summary.py --workload=$N,$B,$DT,$VAR $out/p1 $out/p2 $out/pf $out/pw > $out/pmc_summary.json
find $out -name '*.csv' -size +1M -delete
rm -rf $out/kt $out/p1 $out/p2 $out/pf $out/pw
cat $out/bench.json
python3 - <<PY
import csv, json
for r in csv.DictReader(open("$out/kernel_stats.csv")):
    if 'admpc' in r['Name']: print('%-34s calls %s avg %.1f us' % (r['Name'].split('admpc_')[-1][:32], r['Calls'], float(r['AverageNs'])/1e3))
d = json.load(open("$out/pmc_summary.json"))
for k, v in d.items():
    if not k.startswith('_'): print(k, {c: float('%.4g' % x['mean_per_launch']) for c, x in v.items()})
print(d.get('_step_traffic'))
PY
