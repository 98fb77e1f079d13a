python3 tools/exp_fwd_pairs.py 2:0 0:0 0:4 0:1 0:0
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "pair_kernel or bench_instantiation or macro_rollout_vs or macro_full" 2>&1 | tail -3
