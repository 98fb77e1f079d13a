# checkpoint measurement of HEAD: full GPU suite, smoke, profile round, counter passes, the slow mirror test
P=${1:-r04z}
python -m pytest tests -m gpu -q -s -x > gpurun_out/${P}_gputest.log 2>&1; echo "gputest rc=$?"; tail -3 gpurun_out/${P}_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh $P 2>&1 | tail -40
bash tools/pmc_macro_fwd.sh gpurun_out/${P}_pmc_counters 0 0 > gpurun_out/${P}_pmc_counters.log 2>&1; tail -25 gpurun_out/${P}_pmc_counters.log
DHTS_SLOW=1 timeout 1500 python -m pytest tests/test_itscp_gpu.py -m gpu -q -s -k "matches_reference and hybrid" > gpurun_out/${P}_slow_mirror_test.log 2>&1; echo "slow mirror rc=$?"; tail -4 gpurun_out/${P}_slow_mirror_test.log
