#!/bin/bash
# compiles the C++ host examples against libfpx.so (g++ only: the host layer contains no device code)
set -euo pipefail
cd "$(dirname "$0")"
LINK="-L.. -lfpx -Wl,-rpath,\$ORIGIN/.. -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -pthread"
g++ -O2 -std=c++17 -Wall -o example_search example_search.cpp $LINK
g++ -O2 -std=c++17 -Wall -o test_coalescer test_coalescer.cpp $LINK
g++ -O2 -std=c++17 -Wall -o bench_threads bench_threads.cpp $LINK
echo "built $(realpath example_search) $(realpath test_coalescer) $(realpath bench_threads)"
