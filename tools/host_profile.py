"""cProfile of the host side of bench.py (which Python frames the issue time of a step goes to):
    python tools/host_profile.py <bench.py arguments>     -> top functions by own time"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

sys.argv = ["bench.py"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(28)
