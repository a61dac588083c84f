"""Run ON THE GPU BOX: the light split's taper (ILM_LIGHT_TAPER = f1,f2,f3: the fractions of an XCD's tiles from which on a tile is served by
2, 4 and 8 workgroups; read once per process, hence one process per setting) swept over a grid, on the cost-balanced strips of an
8-rank cfg3 / cfg5 frame rendered one by one on this GPU (tools/strip_probe.py).  Prints max and sum of the eight strips per setting.
    python tools/taper_sweep.py [cfg5|cfg3] [frames]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
frames = sys.argv[2] if len(sys.argv) > 2 else "20"
grid = [None]
for f1 in (0.0, 0.25, 0.4, 0.5, 0.6):
    for f2 in (0.5, 0.65, 0.75, 0.85):
        for f3 in (0.75, 0.875, 0.95, 1.0):
            if f1 <= f2 <= f3:
                grid.append((f1, f2, f3))
for t in grid:
    env = dict(os.environ)
    if t is not None:
        env["ILM_LIGHT_TAPER"] = "%g,%g,%g" % t
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "strip_probe.py"), "--scenes", scene, "--splits", "0", "--frames", frames, "--gbuffer"],
                         env=env, capture_output=True, text=True).stdout
    m = re.search(r"balanced\s+strips:.*?max ([0-9.]+)\s+sum ([0-9.]+)", out)
    w = re.search(r"whole ([0-9.]+) ms", out)
    print("%s taper %-18s whole %s  balanced strips: max %s sum %s" % (scene, "library" if t is None else "%g,%g,%g" % t, w.group(1) if w else "?", m.group(1) if m else "?", m.group(2) if m else "?"), flush=True)
