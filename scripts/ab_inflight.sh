# A/B of bench.py settings on one GPU (ms per step, ms of one event alone): each argument "VAR=val [VAR=val ...] [-- bench args]"
for rep in 1 2; do
for cfg in "$@"; do
  envs="${cfg%%--*}"; args=""; case "$cfg" in *--*) args="${cfg#*--}";; esac
  env $envs python bench.py --no-cpu-baseline --steps 100 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg |', round(d['ms_per_step'],3), round(d['ms_per_fracture_event'],3))"
done; done
