#!/bin/bash
# tools/experiment_isa_stats.sh [patch names ...] -- static figures of the headline probe kernel (k_probe_pgroup<16, BINNED, no per-query
# statistics>) for the product's sources with the named experiments/*.patch applied (none: the product as it is): instructions,
# SGPR spills and the v_readlane / v_writelane traffic they cost, s_nop, scratch, registers.  No GPU needed (hipcc cross-compiles);
# what a variant is WORTH is for tools/probe_ab.py on a GPU to say.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
BIN=/opt/rocm/lib/llvm/bin
K=${FPX_ISA_KERNEL:-_ZN3fpx14k_probe_pgroupILi16ELb1ELb0EEEvNS_9ProbeArgsENS_9GroupArgsE}
T="$(mktemp -d)"
trap 'rm -rf "$T"' EXIT
mkdir -p "$T/acoustid-index_amd"
cp -r "$ROOT/include" "$T/include"
cp -r "$ROOT/acoustid-index_amd/csrc" "$T/acoustid-index_amd/"
for p in "$@"; do (cd "$T" && patch -s -p1 --no-backup-if-mismatch < "$ROOT/experiments/$p.patch"); done
cd "$T/acoustid-index_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result ${FPX_EXTRA_FLAGS:-} -c fpx_search.hip -o "$T/s.o"
$BIN/llvm-objcopy --dump-section=.hip_fatbin="$T/fat.bin" "$T/s.o"
$BIN/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$T/fat.bin" --output="$T/dev.co"
$BIN/llvm-objdump -d --no-show-raw-insn "$T/dev.co" > "$T/all.s"
awk -v k="<$K>:" '$2==k{on=1;next} on&&/^[0-9a-f]+ <_ZN/{exit} on{print}' "$T/all.s" > "$T/k.s"
[ -n "${FPX_ISA_KEEP:-}" ] && cp "$T/k.s" "$FPX_ISA_KEEP"
res=$($BIN/llvm-readelf --notes "$T/dev.co" | awk -v k="$K" '/\.agpr_count:/{blk=""} {blk=blk $0 "\n"} $1==".name:" && $2==k {hit=1} hit && /\.wavefront_size:/{printf "%s", blk; exit}' \
      | grep -E "\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" | tr -s ' ' | tr '\n' ' ' || true)
c() { grep -c "$1" "$T/k.s" || true; }
echo "[${*:-product}] instructions $(grep -c . "$T/k.s")  v_readlane $(c v_readlane)  v_writelane $(c v_writelane)  s_nop $(c s_nop)  scratch $(c scratch_)  saveexec $(c saveexec)  global_load $(c global_load) |$res"
