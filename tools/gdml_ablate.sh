#!/bin/bash
# stage-kernel time at 30 atoms with one phase of the chunk loop compiled out (tools/mkvar.sh abl_X -DGDML_ABLATE_X first)
cd "$(dirname "$0")/.."

for v in "" abl_ROWRED abl_TAIL abl_FORM abl_GRAD abl_MFMA abl_ALL; do
    if [ -n "$v" ]; then export SC_LIB_PATH=$PWD/var/libsc_$v.so; fi
    timeout -k 10 120 python bench.py --config 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())['config5']
print('${v:-full}', d.get('kernels_ms', d))"
done
