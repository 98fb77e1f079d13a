timeout 600 python3 -m pytest tests/test_hybrid_gpu.py -x -q -k "fused_state or short_matches or 600_steps_full" 2>&1 | tail -3
python3 bench.py --workload itscp_hybrid --steps 20 --warmup 3 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hybrid', d['ms_per_step'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})"
python3 tools/probes/exp_hyb_stamps.py
