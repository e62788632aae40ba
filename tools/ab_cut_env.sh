# same-box A/B of the CUT (configs[2]) step under environment switches: bash tools/ab_cut_env.sh "NAME=VAL ..." "NAME=VAL ..."   (first = baseline, run twice)
run() { env $1 python bench.py --model cut --netG segformer_attn_conv --netDs projected_d,basic --proj vitsmall --batch 16 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s' % '$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['config'].get('step_driver'))"; }
for cfg in "$@"; do run "$cfg"; done
run "$1"
