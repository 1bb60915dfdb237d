#!/bin/bash
# A/B library builds for one GPU session: tools/build_variant.sh <name> "<extra flags>" file1 [file2 ...]
# compiles the named sources (relative to intrinsic3d_amd/csrc) with the extra flags and links them with the in-tree objects of everything else into gpurun_ab/lib_<name>.so
# (git-ignored; travels to the GPU box; bench.py / tests load it with I3D_LIB=gpurun_ab/lib_<name>.so).  Run `make -C intrinsic3d_amd/csrc` first.
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/intrinsic3d_amd/csrc
obj=$root/build/obj; var=$root/build/variants/$name; mkdir -p $var $root/gpurun_ab
objs=""
for o in $(find $obj -name '*.o' | sort); do
  rel=${o#$obj/}; src=${rel%.o}; use=$o
  for f in "$@"; do
    if [ "$f" = "$src" ]; then
      fc=""; case $src in device/observe.hip|device/level_kernels.hip|device/mesh_kernels.hip|device/tile_pass_mr.hip|device/fusion_kernels.hip|host/mesh.cpp) fc="-ffp-contract=off";; esac
      x=""; case $src in host/*) x="-x hip";; esac
      mkdir -p $(dirname $var/$rel)
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result $fc $flags $x -c $src -o $var/$rel
      use=$var/$rel
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/gpurun_ab/lib_$name.so $objs -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
echo "built gpurun_ab/lib_$name.so"
