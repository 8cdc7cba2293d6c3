set -e
mkdir -p gpurun_out/r2final
timeout -k 10 120 scripts/probes/vmem_probe > gpurun_out/r2final/vmem_probe.txt 2>&1
echo probe done
timeout -k 10 300 python3 bench.py > gpurun_out/r2final/bench_default.json 2> gpurun_out/r2final/bench_default.err
echo bench default done
timeout -k 10 300 python3 bench.py --batch-per-gpu 8192 --no-cpu-baseline > gpurun_out/r2final/bench_b8192.json 2>> gpurun_out/r2final/bench_default.err
ADMPC_LIB=libadmpc_timers.so timeout -k 10 200 python3 scripts/run_rowqp.py 40 4096 5 > gpurun_out/r2final/timers_n40.log 2>&1
echo timers done
timeout -k 10 500 bash scripts/profile_r2.sh r2final/n40 40 4096 f64 > gpurun_out/r2final/n40_profile.log 2>&1
echo n40 done
timeout -k 10 500 bash scripts/profile_r2.sh r2final/cfg5 80 16384 f32 > gpurun_out/r2final/cfg5_profile.log 2>&1
echo cfg5 done
