#!/bin/bash
# tools/build_variant.sh <name> <extra compiler flags ...> -- acoustid-index_amd/build/exp/libfpx_<name>.so from a COPY of the source tree built
# with the given flags (e.g. -DFPX_PK_PREFETCH=0 -DFPX_PK_WAVES=4): compile-time variants of the kernels for A/B runs on a GPU
# (FPX_LIB=<that library> python tools/probe_ab.py).  The product's build is not touched.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
[ $# -ge 1 ] || { echo "usage: $0 <name> [flags ...]" >&2; exit 2; }
NAME="$1"; shift
T="$(mktemp -d)"
trap 'rm -rf "$T"' EXIT
mkdir -p "$T/acoustid-index_amd" "$ROOT/acoustid-index_amd/build/exp"
cp -r "$ROOT/include" "$T/include"
cp -r "$ROOT/acoustid-index_amd/csrc" "$ROOT/acoustid-index_amd/hostsrc" "$ROOT/acoustid-index_amd/build.sh" "$T/acoustid-index_amd/"
FPX_EXTRA_FLAGS="$*" bash "$T/acoustid-index_amd/build.sh" > /dev/null
cp "$T/acoustid-index_amd/libfpx.so" "$ROOT/acoustid-index_amd/build/exp/libfpx_$NAME.so"
echo "built $ROOT/acoustid-index_amd/build/exp/libfpx_$NAME.so ($*)"
