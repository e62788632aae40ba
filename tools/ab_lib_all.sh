L=joligen_amd/csrc/libjg355.so
cp joligen_amd/csrc/libjg355_A.so /tmp/A.so; cp $L /tmp/B.so
one() { python bench.py "$@" --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], end='  |  ')"; }
for v in A B A B; do cp /tmp/$v.so $L; echo -n "$v  ";
 one --steps 20 --no-cut-leg
 one --model cut --netG segformer_attn_conv --netDs projected_d,basic --proj vitsmall --batch 16 --steps 30 --warmup 5
 one --model cut --netG mobile_resnet_attn --netDs projected_d,basic --batch 16 --steps 20 --warmup 5
 one --model cut --netG resnet --netDs basic --batch 16 --steps 20 --warmup 5
 echo; done
cp /tmp/B.so $L
