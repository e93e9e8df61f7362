import sys, os, cProfile, pstats, threading
sys.argv = ['bench.py', '--no-cpu-baseline', '--rotate-batches', '0', '--no-fg-capped', '--no-profile']
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tools'))
import train_loop as TL
orig = TL.TrainLoop._assemble
prof = cProfile.Profile()
def wrapped(self, idx):
    prof.enable()
    try:
        return orig(self, idx)
    finally:
        prof.disable()
TL.TrainLoop._assemble = wrapped
import bench
try:
    bench.main()
finally:
    pstats.Stats(prof, stream=sys.stderr).sort_stats('tottime').print_stats(14)
