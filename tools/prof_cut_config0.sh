export TMPDIR=/tmp JG_TRACE_MARK=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $O/c0kt -o kt -- python $R/bench.py --model cut --netG resnet --netDs basic --batch 16 --steps 4 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/c0kt.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/c0kt/*.db | head -1) 4 > $O/r06_cut_config0_kernel_stats.md 2>&1
rm -rf $O/c0kt
head -40 $O/r06_cut_config0_kernel_stats.md | cut -c1-150
