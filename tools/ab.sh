#!/bin/bash
# Same-box A/B of two builds of libfosvos_hip.so (GPU box): tools/ab.sh base.so new.so -- <command...>
# runs the command with each library copied into place, A B A B, and leaves the second one installed.
set -e
A=$1; B=$2; shift 3
L=fosvos_amd/lib/libfosvos_hip.so
for r in 1 2; do
  for lib in "$A" "$B"; do
    cp "$lib" $L
    echo "== $lib (round $r)"
    "$@"
  done
done
