python3 tools/exp_fwd_pairs.py 0:0 4:0
for k in f1 f2 f3 w1 w2 w3; do DHTS_LIB=$PWD/diff-hybrid-traffic-sim_amd/csrc/variants/libdhts_$k.so python3 tools/exp_fwd_pairs.py 4:0 4:0; done
