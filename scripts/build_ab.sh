#!/bin/bash
# build_ab.sh name "flags" [name "flags" ...]: variant builds of libvpfx into _ab/ (git-ignored) for scripts/gpu_ab.sh
set -e
cd "$(dirname "$0")/.."
PKG=volumetric-particles-for-unity_amd
rm -rf _ab; mkdir -p _ab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  rm -rf /tmp/ab_obj_$name; mkdir -p /tmp/ab_obj_$name
  make -s -C $PKG/csrc -j8 OBJDIR=/tmp/ab_obj_$name OUT=$PWD/_ab/libvpfx_$name.so EXTRA="$flags" >/dev/null
  echo "built _ab/libvpfx_$name.so ($flags)"
done
