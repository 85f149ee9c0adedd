#!/bin/bash
# compiles the C++ host example against libfpx.so (g++ only: the host layer contains no device code)
set -euo pipefail
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -Wall -o example_search example_search.cpp -L.. -lfpx -Wl,-rpath,'$ORIGIN/..' -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
echo "built $(realpath example_search)"
