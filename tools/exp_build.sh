#!/bin/bash
# exp_build.sh NAME "FLAGS" [unit ...] -- compile an EXPERIMENT build of the library (timing macros; some give wrong results)
# into build/NAME/libciao_hip.so and print that path.  The product library ciaoalgorithms.jl_amd/libciao_hip.so is never
# touched; run a script against the experiment with CIAO_HIP_LIB=<printed path>.  Units named after the flags (e.g. chain_f32
# chain_f64) are NOT recompiled with the flags: their product objects are reused (the flags must not concern them).
set -e
name="$1"; flags="$2"; shift 2 || true
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/build/$name"
for u in "$@"; do
  # api.hip bakes the flags into ciao_build_flags(): it is recompiled for every experiment, whatever the list says -- an
  # experiment library must never report the product library's empty flag string
  [ "$u" = "api" ] && continue
  [ -f "$root/ciaoalgorithms.jl_amd/csrc/$u.o" ] && cp -p "$root/ciaoalgorithms.jl_amd/csrc/$u.o" "$root/build/$name/$u.o" && touch "$root/build/$name/$u.o"
done
make -s -C "$root/ciaoalgorithms.jl_amd/csrc" -j8 EXP="$name" EXTRA="$flags" >/dev/null 2>&1
echo "$root/build/$name/libciao_hip.so"
