#!/bin/bash
# Variant builds of libilluminant_hip.so for on-box A/B runs (tools/step_ab.py, tools/ab_lib.sh): tools/ab/<tag>/libilluminant_hip.so with
# extra compiler flags for ONE translation unit.     tools/ab_build.sh <tag> <file.hip> [flags...]      (tag "base": a copy of the shipped library)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/illuminant_amd/csrc"
tag=$1; shift
mkdir -p "$ROOT/tools/ab/$tag"
if [ "$tag" = base ]; then make -s -j4; cp ../lib/libilluminant_hip.so "$ROOT/tools/ab/base/"; exit 0; fi
src=$1; shift
objs=""
for o in particles lighting fields gbuffer output raster api group; do
  if [ "$o.hip" = "$src" ]; then
    case $o in particles|lighting|fields|raster) noslp=-fno-slp-vectorize;; *) noslp=;; esac      # as the Makefile
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $noslp "$@" -c $src -o /tmp/ab_${tag}_$o.o
    objs="$objs /tmp/ab_${tag}_$o.o"
  else objs="$objs $o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/$tag/libilluminant_hip.so" $objs -ldl
