# same-box A/B of two library builds on the UNet legs (palette C2, C4 512x512, cm): bash tools/ab_lib_unet.sh   (A = joligen_amd/csrc/libjg355_A.so, B = the built library)
L=joligen_amd/csrc/libjg355.so
cp joligen_amd/csrc/libjg355_A.so /tmp/A.so; cp $L /tmp/B.so
one() { python bench.py "$@" --no-cpu-baseline --no-kernel-timing --no-cut-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], end='  |  ')"; }
for v in A B A B; do cp /tmp/$v.so $L; echo -n "$v  ";
 one --steps 30
 one --size 512 --batch 8 --efficient 0 --steps 20
 one --model cm --batch 64 --steps 10
 echo; done
cp /tmp/B.so $L
