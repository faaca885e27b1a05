#!/bin/bash
# Builds variants/libmcull_<name>.so from the tree with extra compiler flags (here, on the CPU; hipcc cross-compiles):
#   tools/build_variant.sh <name> [-DMACRO=value ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
B=/tmp/mcull_variant_$NAME
rm -rf $B && mkdir -p $B/pkg/csrc $B/include $R/variants
cp $R/vulkan-renderer_amd/csrc/*.hip $R/vulkan-renderer_amd/csrc/*.h $R/vulkan-renderer_amd/csrc/Makefile $B/pkg/csrc/
cp $R/include/*.h $B/include/
make -s -C $B/pkg/csrc -j8 FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter $*" OUT=$R/variants/libmcull_$NAME.so
ls -la $R/variants/libmcull_$NAME.so
