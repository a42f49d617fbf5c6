#!/bin/bash
# A/B: rebuild the HIP library with extra -D flags and run a command for each variant.
# usage: tools/ab.sh "<cmd>" "" "-DX" "-DY -DZ"
cmd="$1"; shift
export MGL_NO_AUTOBUILD=1
for fl in "$@"; do
  echo "=== variant: [$fl]"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $fl -o megalania_amd/_build/libmegalania_hip.so megalania_amd/csrc/mgl_api.hip || exit 1
  bash -c "$cmd" || exit 1
done
