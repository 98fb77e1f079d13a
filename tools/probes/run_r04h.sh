python3 tools/exp_fwd_pairs.py 0:0 4:0
for k in fair3 fair5 fair7 fair9; do DHTS_LIB=$PWD/diff-hybrid-traffic-sim_amd/csrc/variants/libdhts_$k.so python3 tools/exp_fwd_pairs.py 4:0 4:0; done
python3 tools/exp_fwd_pairs.py 4:0
