export TMPDIR=/tmp JG_TRACE_MARK=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/gkt -o kt -- python $R/bench.py --model cut --netG segformer_attn_conv --netDs projected_d,basic --proj vitsmall --batch 16 --steps 4 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/gkt.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/gkt/*.db | head -1) 4 --by-grid > $O/r06_cut_kernel_stats_by_grid.md 2>&1
rm -rf $O/gkt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/gkt -o kt -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-cut-leg > $O/gkt.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/gkt/*.db | head -1) 4 --by-grid > $O/r06_kernel_stats_by_grid.md 2>&1
rm -rf $O/gkt
