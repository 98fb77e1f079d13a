# One measurement of HEAD on the GPU box: full GPU suite (-s), smoke(), profile round, counter passes.  Usage: bash tools/measure_round.sh <prefix>; then tools/collect_profiles.sh <prefix>
P=${1:-r04z}
python -m pytest tests -m gpu -q -s > gpurun_out/${P}_gputest.log 2>&1; echo "gputest rc=$?"; tail -3 gpurun_out/${P}_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh $P 2>&1 | tail -40
bash tools/pmc_workloads.sh gpurun_out/${P}_pmc_counters > gpurun_out/${P}_pmc_counters.log 2>&1; tail -25 gpurun_out/${P}_pmc_counters.log
