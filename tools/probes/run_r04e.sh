python3 tools/exp_fwd_pairs.py 0:0 3:0 4:0 4:4 3:0 4:0
python3 tools/probes/exp_fwd3_stamps.py 4 0 1024
