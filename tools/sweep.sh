#!/bin/bash
# Same-box sweep of configurations of the online-step bench (GPU box).  Each argument is "label|lib.so|ENV=val ENV2=val";
# all configurations run round-robin ROUNDS times (default 2), one line per run: label, frames/s.
#   tools/sweep.sh "base|build/ab/a.so|" "blocks160|build/ab/a.so|FOSVOS_WGRAD_BLOCKS=160"
ROUNDS=${ROUNDS:-2}
STEPS=${STEPS:-100}
L=fosvos_amd/lib/libfosvos_hip.so
cp $L /tmp/sweep_keep.so
for r in $(seq $ROUNDS); do
  for cfg in "$@"; do
    IFS='|' read -r label lib envs <<< "$cfg"
    [ -n "$lib" ] && cp "$lib" $L
    fps=$(env $envs timeout -k 10 200 python bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | python -c "import json,sys; print('%.1f' % json.loads(sys.stdin.readline())['value'])")
    echo "$label r$r $fps"
  done
done
cp /tmp/sweep_keep.so $L
