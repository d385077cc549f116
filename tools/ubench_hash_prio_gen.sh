#!/bin/bash
# Generates six block variants of tools/ubench_hash_yield.hip from full hashgen.py option strings and compiles it (round 5:
# runs by issue class + wave priority).  usage: bash tools/ubench_hash_prio_gen.sh "opts0" .. "opts5"   (each: hashgen.py options)
cd "$(dirname "$0")"
: > hb_labels.h
for i in 0 1 2 3 4 5; do
  eval "O=\${$((i+1))}"
  python3 ../vgen_amd/csrc/device/hashgen.py $O > hb_$i.inc
  echo "#define HB_LABEL_$i \"block $O\"" >> hb_labels.h
  echo "#define HB_CALL_$i(p, x, h, f) hb$i::hash160_pub33_block(p, x, h)" >> hb_labels.h
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed ubench_hash_yield.hip -o ubench_hash_yield
