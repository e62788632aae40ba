# same-box A/B of the consistency-model step (configs[4] shape) under environment switches: bash tools/ab_cm_env.sh "NAME=VAL" ...
run() { env $1 python bench.py --model cm --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-cut-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for cfg in "$@"; do run "$cfg"; done
run "$1"
