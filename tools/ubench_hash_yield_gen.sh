#!/bin/bash
# Generates the six block variants of tools/ubench_hash_yield.hip and compiles it.
# usage: bash tools/ubench_hash_yield_gen.sh [mode0 .. mode5]      (hashgen.py --yield modes; default below)
cd "$(dirname "$0")"
MODES=("${@:-none every:3 every:2 every:4 every:3:salu every:4:salu}")
[ $# -eq 0 ] && MODES=(none every:3 every:2 every:4 every:3:salu every:4:salu)
: > hb_labels.h
for i in 0 1 2 3 4 5; do
  M="${MODES[$i]}"; INSN="s_nop 0"; FILL=""
  case "$M" in *"+"*) FILL="--filler ${M#*+}"; M="${M%%+*}";; esac   # "mode+seq:N" / "mode+mix:N": N multiply-adds after / spread through the block
  case "$M" in *"|"*) INSN="${M#*|}"; M="${M%%|*}";; esac            # "mode|instruction": another yield instruction
  python3 ../vgen_amd/csrc/device/hashgen.py --class-window 0 --prio none --yield "$M" --yield-insn "$INSN" $FILL > hb_$i.inc   # (round 4's dependency order: the yields' own A/B)
  echo "#define HB_LABEL_$i \"block ${MODES[$i]}\"" >> hb_labels.h
  if [ -n "$FILL" ]; then echo "#define HB_CALL_$i(p, x, h, f) hb$i::hash160_pub33_block(p, x, f, h)" >> hb_labels.h
  else echo "#define HB_CALL_$i(p, x, h, f) hb$i::hash160_pub33_block(p, x, h)" >> hb_labels.h; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed ubench_hash_yield.hip -o ubench_hash_yield
