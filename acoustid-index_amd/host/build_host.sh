#!/bin/bash
# compiles the C++ host examples against libfpx.so (g++ only: the host layer contains no device code).
# Each binary is linked under a private name and renamed into place: several callers may build at once (the variant runs of the
# test suite), and a program that is being executed must not be written to.
set -euo pipefail
cd "$(dirname "$0")"
LINK="-L.. -lfpx -Wl,-rpath,\$ORIGIN/.. -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -pthread"
for prog in example_search test_coalescer bench_threads; do
  if [ -x $prog ] && [ $prog -nt $prog.cpp ] && [ $prog -nt fpx.hpp ] && [ $prog -nt fpx_coalescer.hpp ] && [ $prog -nt ../../include/fpx.h ]; then continue; fi
  g++ -O2 -std=c++17 -Wall -o .$prog.$$ $prog.cpp $LINK
  mv -f .$prog.$$ $prog
done
echo "built $(realpath example_search) $(realpath test_coalescer) $(realpath bench_threads)"
