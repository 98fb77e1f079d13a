cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04v_prof -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --workload itscp_hybrid > /root/repo/gpurun_out/r04v_bench.json 2> /root/repo/gpurun_out/r04v_bench.err
S=$(find /root/repo/gpurun_out/r04v_prof -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $S | cut -c1-110 | head -16
tail -c 300 /root/repo/gpurun_out/r04v_bench.json
