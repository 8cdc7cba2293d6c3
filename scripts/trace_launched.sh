#!/bin/bash
# kernel timeline of the one-rank launched bench (arg-min side stream active): where do the microseconds between two solves go?
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
out=gpurun_out/trl; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trl/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
sel = rows[-60:]
for r in sel:
    print("%10.1f %8.1f us  q%-3s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
PY
