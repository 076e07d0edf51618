#!/bin/bash
# Diagnostic: build an experimental variant of libcvae_hip.so with extra compile flags for ONE source file.
# usage: scripts/build_variant.sh NAME FILE.hip "-DFOO=1 ..."   ->  coupled-vae_amd/lib/var/libcvae_hip_NAME.so
# Select it at run time with CVAE_HIP_LIB=<that path>.
set -e
cd "$(dirname "$0")/../coupled-vae_amd/csrc"
make -s
name=$1; file=$2; flags=$3
case "$flags" in *_EXP_*) flags="$flags -DCVAE_TIMING_BUILD";; esac      # results are wrong: cvae_version() < 0
mkdir -p build/var ../lib/var
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c "$file" -o "build/var/${name}.o"
objs=""
for o in build/*.o; do
  if [ "$o" = "build/${file%.hip}.o" ]; then objs="$objs build/var/${name}.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "../lib/var/libcvae_hip_${name}.so" $objs
echo "built coupled-vae_amd/lib/var/libcvae_hip_${name}.so"
