python3 tools/exp_fwd_pairs.py 0:0 4:0 5:0 5:4 4:4 5:0
python3 tools/probes/exp_fwd3_stamps.py 5 0 1024
