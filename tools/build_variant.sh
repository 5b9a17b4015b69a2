#!/bin/bash
# A/B library with ONE source compiled under extra flags:   tools/build_variant.sh <name> <source.hip> <flags...>
# -> build/<name>/libnlstack.so (every other object is the default build's; run with NLSTACK_LIB=$PWD/build/<name>/libnlstack.so)
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/nightlight_amd/csrc
mkdir -p $root/build/$name
make -s -C $cs -j8
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
(cd $cs && /opt/rocm/bin/hipcc $F "$@" -c $src -o $root/build/$name/${src%.hip}.o)
objs=$(cd $cs && ls *.o | grep -v '\.exp\.o' | grep -v "^${src%.hip}\.o$" | sed "s|^|$cs/|")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $root/build/$name/${src%.hip}.o -o $root/build/$name/libnlstack.so -lpthread
echo built $root/build/$name/libnlstack.so
