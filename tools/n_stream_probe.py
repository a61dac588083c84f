"""Experiment: the cfg2 step over N HIP streams (N contexts on one GPU, the library's own two-stream split switched off with
ILM_STEP_STREAMS=1): is there anything past two?      ILM_STEP_STREAMS=1 python tools/n_stream_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("ILM_STEP_STREAMS", "1")
from two_stream_probe import make, run, native      # (runs that probe's own cases first)

for split in ((16,), (8, 8), (6, 5, 5), (4, 4, 4, 4), (3, 3, 3, 3, 2, 2)):
    parts = []
    for i, n in enumerate(split):
        c = native.Context(0)
        e, s, d = make(c, 256, n, 10 + 40 * i, i == 0)
        parts.append((c, s, d, e))
    print("%d stream(s), chunks %s (+1 spawn target on the first): %.2f us/step" % (len(split), " | ".join(str(n) for n in split),
                                                                                  run([(c, s, d) for (c, s, d, e) in parts])))
