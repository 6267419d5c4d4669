#!/bin/bash
# usage: tools/build_variant.sh NAME PATCH.py   -- builds bpmf_amd/csrc/variants/NAME.so from a copy of
# csrc/ patched by PATCH.py (run with the copy's path as argv[1]); the working tree is left untouched
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/variant_$1; rm -rf $W; mkdir -p $W/bpmf_amd; cp -r $ROOT/bpmf_amd/csrc $W/bpmf_amd/csrc; cp -r $ROOT/include $W/include
rm -f $W/bpmf_amd/csrc/*.o
python $2 $W/bpmf_amd/csrc
make -s -C $W/bpmf_amd/csrc $W/bpmf_amd/csrc/../libbpmf_hip.so 2>&1 | grep -E " error|error:" -A5 || true
mkdir -p $ROOT/bpmf_amd/csrc/variants
cp $W/bpmf_amd/libbpmf_hip.so $ROOT/bpmf_amd/csrc/variants/$1.so
ls -la $ROOT/bpmf_amd/csrc/variants/$1.so
