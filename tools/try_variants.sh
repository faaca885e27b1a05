#!/bin/bash
# Measures library variants built under variants/ (libmcull_<name>.so) with one bench command each, on the GPU box:
#   tools/try_variants.sh "<bench.py arguments>" base name1 name2 ...
# "base" is the library of the tree.  Each run has its own time limit; a failing run ends the script.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ARGS=$1; shift
cp vulkan-renderer_amd/libmcull.so /tmp/libmcull_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libmcull_base.so vulkan-renderer_amd/libmcull.so; else cp variants/libmcull_$v.so vulkan-renderer_amd/libmcull.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline $ARGS > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || { tail -5 gpurun_out/var_$v.err; exit 1; }
  python3 -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[2], j['ms_per_step'], 'ms; kernel_ms', j['roofline'].get('kernel_ms'))" gpurun_out/var_$v.json $v
done
cp /tmp/libmcull_base.so vulkan-renderer_amd/libmcull.so
