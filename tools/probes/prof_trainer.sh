cd diff-hybrid-traffic-sim_amd
python -c "
import cProfile, pstats, sys, io
sys.argv = ['run', '--mode=hybrid', '--n_trial=1', '--n_intersection=3', '--n_lane=1', '--lane_length=5', '--simulation_length=20', '--signal_length=4', '--n_episode=30', '--lr=1e-4']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_module('example.control.itscp.run', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35)
print(s.getvalue()[:6000])
" 2>&1 | tail -60
