#!/bin/bash
# Placement sweep of the headline kernel: every configuration in fresh processes (REPS each).  Output: $1 (a text file of JSON lines).
OUT=${1:-gpurun_out/headline_layout.txt}
REPS=${REPS:-3}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
run() { for i in $(seq "$REPS"); do python3 tools/headline_layout.py "$@" >> "$OUT" 2>> "$OUT.err" || echo "{\"failed\": \"$*\"}" >> "$OUT"; done; }
run --label one_pad0
run --label separate_pad0 --mode separate
run --label one_pad64 --pad 64
run --label stagger4k --mode stagger --stagger-bytes 4096
run --label stagger64k --mode stagger --stagger-bytes 65536
run --label stagger2M+4k --mode stagger --stagger-bytes 2101248
run --label one_pad0_noswizzle --tune 256,2,1,0
run --label one_pad0_depth3 --tune 256,3,1,1
run --label one_pad0_b512 --tune 512,2,1,1
