# same-box A/B of other CUT selections under environment switches: bash tools/ab_cut0_env.sh "<netG> <netDs>" "NAME=VAL ..." "NAME=VAL ..."   (first setting = baseline, run twice)
SEL=$1; shift
set -- $SEL "$@"; G=$1; D=$2; shift 2
run() { env $1 python bench.py --model cut --netG $G --netDs $D --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s' % '$1', '$G', '$D', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['config'].get('step_driver'))"; }
for cfg in "$@"; do run "$cfg"; done
run "$1"
