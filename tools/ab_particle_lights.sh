#!/bin/bash
# Run ON THE GPU BOX: tools/particle_lights_ab.py with variant builds of the library (tools/ab/<tag>/).  tools/ab_particle_lights.sh w6 w7 ...
cd "$(cd "$(dirname "$0")/.." && pwd)"
for tag in "$@"; do
  echo "== $tag"
  ILM_HIP_LIB=$PWD/tools/ab/$tag/libilluminant_hip.so LD_LIBRARY_PATH=$PWD/tools/ab/$tag:${LD_LIBRARY_PATH:-} python tools/particle_lights_ab.py 3 10 2>&1 | grep -v "^stats\|statistics"
done
