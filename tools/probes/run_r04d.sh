python3 tools/probes/exp_fwd3_stamps.py 0 0 1024 | grep -v 'wave\|phase 1\|cycles per'
python3 tools/probes/exp_fwd3_stamps.py 3 0 1024
python3 tools/probes/exp_fwd3_stamps.py 3 0 512 | grep -v 'wave\|phase 1\|cycles per'
