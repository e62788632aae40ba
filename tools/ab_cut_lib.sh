# same-box A/B of a CUT selection between two builds of the library: bash tools/ab_cut_lib.sh "<netG> <netDs> [--proj X]" A.so [B.so = the built one]   (runs A B A B)
SEL=$1; L=joligen_amd/csrc/libjg355.so
cp ${2:?path of build A} /tmp/A.so; cp ${3:-$L} /tmp/B.so
set -- $SEL; G=$1; D=$2; shift 2
for v in A B A B; do cp /tmp/$v.so $L; python bench.py --model cut --netG $G --netDs $D "$@" --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$G', '$D', d['value'], d['ms_per_step'], d['config'].get('step_driver'))"; done
cp /tmp/B.so $L
