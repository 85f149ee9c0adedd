#!/bin/bash
# tools/build_experiment.sh <name> [more patch names ...] -- builds acoustid-index_amd/build/exp/libfpx_<name>.so from a COPY of the
# source tree with experiments/<name>.patch (and the others named) applied: kernel variants for A/B measurements on a GPU
# (FPX_LIB=<that library> python tools/probe_ab.py; tools/ab.sh).  The product's sources are not touched -- bench.py's
# kernel_source_sha16 keeps naming the kernels the committed profiles were measured on -- until a variant has been measured,
# has passed `FPX_LIB=... pytest -m gpu`, and is merged into csrc/ for good.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
[ $# -ge 1 ] || { echo "usage: $0 <patch name> [...]" >&2; exit 2; }
NAME="$(IFS=+; echo "$*")"
T="$(mktemp -d)"
trap 'rm -rf "$T"' EXIT
mkdir -p "$T/acoustid-index_amd" "$ROOT/acoustid-index_amd/build/exp"
cp -r "$ROOT/include" "$T/include"
cp -r "$ROOT/acoustid-index_amd/csrc" "$ROOT/acoustid-index_amd/hostsrc" "$ROOT/acoustid-index_amd/build.sh" "$T/acoustid-index_amd/"
for p in "$@"; do
  (cd "$T" && patch -p1 --no-backup-if-mismatch < "$ROOT/experiments/$p.patch")
done
FPX_EXTRA_FLAGS="${FPX_EXTRA_FLAGS:-}" bash "$T/acoustid-index_amd/build.sh" > /dev/null
cp "$T/acoustid-index_amd/libfpx.so" "$ROOT/acoustid-index_amd/build/exp/libfpx_$NAME.so"
echo "built $ROOT/acoustid-index_amd/build/exp/libfpx_$NAME.so"
