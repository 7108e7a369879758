"""Profiling helper (not a test): where the HOST time of the training frame step goes (cProfile over 40 frames of
tools/train_throughput.py's loop; Event.synchronize is the wait for the valid-pixel count)."""
import cProfile, pstats, io, os, sys, runpy
sys.argv = [os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_throughput.py')]
pr = cProfile.Profile()
pr.enable()
runpy.run_path(sys.argv[0], run_name='__main__')
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumtime').print_stats(40)
print(s.getvalue()[:8000])
