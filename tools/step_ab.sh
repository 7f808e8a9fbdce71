#!/bin/bash
# In-step A/B on the GPU box: tools/step_ab.sh <out_dir> <rounds> name1=ENV1=V,ENV2=V[@lib.so] name2=... ; runs bench.py (no CPU
# baseline, no inference, no variants) once per variant and round, interleaved, and prints frames/s per run.
OUT=$1; ROUNDS=$2; shift 2
L=fosvos_amd/lib/libfosvos_hip.so
cp $L /tmp/lib_default.so
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%=*}; rest=${v#*=}
    lib=/tmp/lib_default.so
    if [[ "$rest" == *@* ]]; then lib=${rest##*@}; rest=${rest%@*}; fi
    cp $lib $L
    envs=$(echo "$rest" | tr ',' ' ')
    [ "$envs" == "-" ] && envs="FOO=1"
    env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-infer --no-variants > $OUT/${name}_$r.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("$OUT/${name}_$r.json"))
k=d["roofline"]["by_kernel"]
def t(p): return sum(v["ms_per_step"] for n,v in k.items() if n.startswith(p))
print(f"$name round $r: {d['value']:.1f} frames/s  igemm {t('k_conv3x3_igemm'):.4f} wgrad {t('k_wgrad3x3'):.4f} all {d['roofline']['device_ms_per_step_all_kernels']:.4f} ms/step")
PY
  done
done
cp /tmp/lib_default.so $L
