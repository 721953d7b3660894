#!/bin/bash
# Builds librubiks_hip.so as of a git revision (or of the working tree: rev = WORK) into rl-rubiks_amd/lib/ab/<name>.so, for same-box
# A/B runs of kernel changes (RUBIKS_HIP_LIB=<that file> selects it; the ABI must be the one of the working tree's Python side).
#   tools/build_ab_lib.sh <git-rev|WORK> <name> [extra compiler flags, e.g. -DRUBIKS_GEMM_ABLATE=1]
set -e
rev=$1; name=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
if [ "$rev" = WORK ]; then
    mkdir -p "$tmp/rl-rubiks_amd"
    cp -r "$root/rl-rubiks_amd/csrc" "$root/rl-rubiks_amd/Makefile" "$tmp/rl-rubiks_amd/"
    cp -r "$root/include" "$tmp/"
else
    git -C "$root" archive "$rev" rl-rubiks_amd/csrc rl-rubiks_amd/Makefile include | tar -x -C "$tmp"
fi
make -C "$tmp/rl-rubiks_amd" -j8 EXTRA="$extra" > /dev/null
mkdir -p "$root/rl-rubiks_amd/lib/ab"
cp "$tmp/rl-rubiks_amd/lib/librubiks_hip.so" "$root/rl-rubiks_amd/lib/ab/$name.so"
rm -rf "$tmp"
echo "$root/rl-rubiks_amd/lib/ab/$name.so"
