#!/bin/bash
# A/B build of libgeoac_hip.so beside the shipped one: usage  tools/build_ab.sh <name> [extra hipcc flags, e.g. -DGEOAC_X=0] [AB=1]
# -> build_ab_<name>/libgeoac_hip.so (git-ignored, travels with gpurun).  AB=1 as a second argument word adds the diagnostic kernels (k_rk4_duo, the grid
# sets' two-lane kernels).  Use with tools/ab_metric.py <passes> <lib> <lib> ... or GEOAC_LIB=<lib> python -m pytest tests -m gpu -k ...
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
AB=""; EXTRA=""
for a in "$@"; do case "$a" in AB=1) AB="AB=1" ;; *) EXTRA="$EXTRA $a" ;; esac; done
mkdir -p "$HERE/build_ab_$NAME"
make -s -C "$HERE/geoac_amd/csrc" ARCH=gfx950 $AB EXTRA="$EXTRA" OUT="$HERE/build_ab_$NAME/libgeoac_hip.so" OBJDIR="$HERE/build_ab_$NAME/obj" "$HERE/build_ab_$NAME/libgeoac_hip.so"
ls -la "$HERE/build_ab_$NAME/libgeoac_hip.so"
