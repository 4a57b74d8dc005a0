#!/bin/bash
# build the current tree as a TUNING library (-DSC_TUNING: environment knobs, rejected kernel variants) for A/B runs:  tools/mkvar.sh NAME [extra hipcc flags]   -> var/libsc_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p var
python -m semiclassical_amd.build --out var/libsc_$name.so -DSC_TUNING "$@" | tail -1
