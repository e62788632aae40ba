# same-box A/B of the palette step between two builds of the library: bash tools/ab_lib.sh joligen_amd/csrc/libjg355_A.so [B.so = the built libjg355.so]
# (runs A B A B, 30 timed steps each; the built library is restored at the end)
L=joligen_amd/csrc/libjg355.so
cp ${1:?path of build A} /tmp/A.so; cp ${2:-$L} /tmp/B.so
for v in A B A B; do cp /tmp/$v.so $L; python bench.py --steps 30 --no-cpu-baseline --no-cut-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; done
cp /tmp/B.so $L
