"""Experiment: how long does a launch take that only spawns (cfg2's record: 1092 slots, spherical position + velocity formulas) -- i.e. the
latency of the spawning waves alone -- against an update-only launch of the same nearly empty chunk?  HIP-event timing of back-to-back launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("tsl", os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_stream_lib.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from illuminant_amd import native
a = native.Context(0)
for spawn in (True, False):
    e, s, d = m.make(a, 256, 0 if spawn else 1, 10, spawn)
    if not spawn:
        d.ChunkCount = 1
    print("chunks=1 spawn=%d: %.2f us/step" % (spawn, m.run([(a, s, d)], 300)))
