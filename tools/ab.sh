#!/bin/bash
# time variant libraries back to back on the GPU box:  tools/ab.sh NAME...   (after tools/mkvar.sh NAME)
cd "$(dirname "$0")/.."
for v in "$@"; do
    echo "== $v"
    SC_LIB_PATH=$PWD/var/libsc_$v.so OCC_LIST=${OCC_LIST:-4} timeout -k 10 120 python tools/phase_timing.py ${N:-100000} 2>&1 | grep occ= || exit 1
done
