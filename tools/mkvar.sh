#!/bin/bash
# build the current tree as a TUNING library (-DSC_TUNING: environment knobs, rejected kernel variants) for A/B runs:  tools/mkvar.sh NAME [extra hipcc flags]   -> var/libsc_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -I semiclassical_amd/csrc \
    -DSC_TUNING -o var/libsc_$name.so "$@" semiclassical_amd/csrc/*.hip tools/variants/*.hip
echo var/libsc_$name.so
