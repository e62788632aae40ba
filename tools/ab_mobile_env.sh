run() { env $1 python bench.py --model cut --netG mobile_resnet_attn --netDs projected_d,basic --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-20s' % '$1', d['value'], d['ms_per_step'])"; }
for c in "JG_GEMM_TALL=0" "JG_GEMM_TALL=1" "JG_GEMM_TALL=0" "JG_GEMM_TALL=1"; do run $c; done
