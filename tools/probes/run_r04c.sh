set -x
python3 tools/probes/exp_fwd3_stamps.py 3 0 512
for k in 1 2 3; do DHTS_LIB=$PWD/diff-hybrid-traffic-sim_amd/csrc/variants/libdhts_skew$k.so python3 tools/exp_fwd_pairs.py 3:0 3:0; done
python3 tools/exp_fwd_pairs.py 3:0
