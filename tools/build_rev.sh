#!/bin/bash
# measurement aid (build container): the library as of git revision REV built beside the current one, as
# space_gym_amd/lib/libspacegym_hip_SUFFIX.so, for same-box A/B runs with tools/gpu_ab.sh
#   tools/build_rev.sh REV SUFFIX [extra hipcc flags]
set -e
rev=$1; suf=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" space_gym_amd/csrc include | tar -x -C "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -ffp-contract=on -I "$tmp/include" -I "$tmp/space_gym_amd/csrc" "$@" \
  -o "$root/space_gym_amd/lib/libspacegym_hip_$suf.so" "$tmp/space_gym_amd/csrc/sg_engine.hip"
rm -rf "$tmp"
echo "$root/space_gym_amd/lib/libspacegym_hip_$suf.so"
