python3 tools/exp_hyb_variants.py hstamps
python3 tools/probes/exp_hyb_stamps.py
DHTS_LIB=diff-hybrid-traffic-sim_amd/csrc/variants/libdhts_hstamps.so python3 tools/probes/exp_hyb_trace.py
timeout 900 python -m pytest tests/test_hybrid_gpu.py tests/test_itscp_gpu.py -m gpu -q -x 2>&1 | tail -3
