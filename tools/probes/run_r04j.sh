python3 tools/exp_fwd_pairs.py
python3 tools/probes/exp_fwd3_stamps.py 0 0 1024
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "macro" 2>&1 | tail -15
