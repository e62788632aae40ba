# same-box A/B of the palette step under environment switches: bash tools/ab_env.sh "NAME=VAL ..." "NAME=VAL ..." ...   (first = baseline, run twice)
run() { env $1 python bench.py --steps 30 --no-cpu-baseline --no-cut-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s' % '$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for cfg in "$@"; do run "$cfg"; done
run "$1"
