V=diff-hybrid-traffic-sim_amd/csrc/variants
for n in base512 split512; do
  for L in 512 1024; do
    echo "== $n lanes $L"
    DHTS_LIB=$V/libdhts_$n.so DHTS_EXP_LANES=$L python3 tools/exp_fwd_pairs.py 0:2 0:4 0:1 0:2
  done
done
