#!/bin/bash
# Builds libfpx.so (HIP, gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")/csrc"
OUT=../libfpx.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result ${FPX_EXTRA_FLAGS:-}"      # (FPX_EXTRA_FLAGS: A/B builds, e.g. -DFPX_FK_WG=512)
mkdir -p ../build
pids=()
stale() {      # $1: source, $2: object
  [ ! -f $2 ] || [ $1 -nt $2 ] || [ fpx_internal.h -nt $2 ] || [ -n "$(find . -maxdepth 1 -name "*.hpp" -newer $2 2>/dev/null)" ] || [ ../../include/fpx.h -nt $2 ]
}
for f in fpx_sort fpx_search fpx_api fpx_build fpx_group fpx_sharded; do
  if stale $f.hip ../build/$f.o; then
    hipcc $FLAGS -c $f.hip -o ../build/$f.o &
    pids+=($!)
  fi
done
# hostsrc/: entry points that are host code over the other entry points (no kernel of their own; csrc/ is the kernels' tree,
# whose hash bench.py prints next to every measurement)
for f in fpx_hist; do
  if stale ../hostsrc/$f.hip ../build/$f.o; then
    hipcc $FLAGS -I. -c ../hostsrc/$f.hip -o ../build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../build/fpx_sort.o ../build/fpx_search.o ../build/fpx_api.o ../build/fpx_build.o ../build/fpx_group.o ../build/fpx_sharded.o ../build/fpx_hist.o -lpthread
echo "built $(realpath $OUT)"
