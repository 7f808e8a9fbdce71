# actual GFX clocks / socket power (amd-smi, read-only) once a second while a command runs:  smi_probe.sh <command ...>
# default: the fine-tune step for 6000 iterations
if [ $# -eq 0 ]; then set -- python bench.py --steps 6000 --warmup 50 --no-cpu-baseline --no-infer --no-variants --no-alone --no-roofline; fi
"$@" > /tmp/smi_probe_cmd.out 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  amd-smi metric --clock --power --usage 2>/dev/null | grep -E "GFX_ACTIVITY|SOCKET_POWER|^ +GFX_[0-7]:|^ +CLK:" | tr -s ' ' | tr '\n' ' ' | cut -c1-330
  echo
  sleep 1
done
wait $BP
tail -12 /tmp/smi_probe_cmd.out | cut -c1-200
