#!/bin/bash
# exp_build.sh NAME "FLAGS" -- compile an EXPERIMENT build of the library (timing macros; some give wrong results) into
# build/NAME/libciao_hip.so and print that path.  The product library ciaoalgorithms.jl_amd/libciao_hip.so is never
# touched; run a script against the experiment with CIAO_HIP_LIB=<printed path>.
set -e
name="$1"; flags="$2"
root="$(cd "$(dirname "$0")/.." && pwd)"
make -s -C "$root/ciaoalgorithms.jl_amd/csrc" -j8 EXP="$name" EXTRA="$flags" >/dev/null 2>&1
echo "$root/build/$name/libciao_hip.so"
