V=diff-hybrid-traffic-sim_amd/csrc/variants
for n in "" fair4a fair4b fair4c ""; do
  echo "== ${n:-product}"
  if [ -z "$n" ]; then python3 tools/exp_fwd_pairs.py 0:0 0:0 | cut -c1-100; else DHTS_LIB=$V/libdhts_$n.so python3 tools/exp_fwd_pairs.py 0:0 0:0 | cut -c1-100; fi
done
