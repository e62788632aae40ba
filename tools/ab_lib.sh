cd joligen_amd/csrc; cp libjg355.so /tmp/A.so; cp libjg355_noslp.so /tmp/B.so; cd ../..
for v in A B A B; do cp /tmp/$v.so joligen_amd/csrc/libjg355.so; python bench.py --steps 30 --no-cpu-baseline --no-cut-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; done
cp /tmp/A.so joligen_amd/csrc/libjg355.so
