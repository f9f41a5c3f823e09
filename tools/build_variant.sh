#!/bin/bash
# An experiment build of the library next to the product, for A/B runs on one GPU box (tools/exp_variants.py):
#   tools/build_variant.sh <name> "<extra flags>" [translation units to recompile ...]
# -> rust-eth-kzg_amd/ab/libc_eth_kzg_<name>.so (git-ignored; travels to the GPU box with the snapshot).  With a list of translation
# units only those are recompiled with the flags (the other objects are the product's); without one, everything is.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$REPO/rust-eth-kzg_amd/csrc
NAME=$1; FLAGS=$2; shift 2
OBJ=$CSRC/build_$NAME
mkdir -p "$REPO/rust-eth-kzg_amd/ab" "$OBJ"
if [ $# -gt 0 ]; then
  make -C "$CSRC" -j8 HOOKS=0 > /dev/null   # the product's objects, up to date
  cp -p "$CSRC"/build/*.o "$OBJ"/
  for tu in "$@"; do rm -f "$OBJ/$tu.o"; done
  # only the removed objects are rebuilt: the copies are as new as the product's
fi
make -C "$CSRC" -j8 HOOKS=0 EXTRA="$FLAGS" OBJDIR="$OBJ" OUT="$REPO/rust-eth-kzg_amd/ab/libc_eth_kzg_$NAME.so" STATIC="$OBJ/unused.a" 2>&1 | grep -E "error|warning: unused|Error" || true
ls -la "$REPO/rust-eth-kzg_amd/ab/libc_eth_kzg_$NAME.so"
