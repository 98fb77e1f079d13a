python3 tools/exp_fwd_pairs.py 0:0 4:0 4:4 5:0 4:0 4:4
python3 tools/probes/exp_fwd3_stamps.py 4 0 1024
