# kernel trace of another CUT selection: bash tools/prof_cut_sel.sh <netG> <netDs> <out name>   (GPU box; -> gpurun_out/<out name>)
export TMPDIR=/tmp JG_TRACE_MARK=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/selkt -o kt -- python $R/bench.py --model cut --netG $1 --netDs $2 --batch 16 --steps 4 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/selkt.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/selkt/*.db | head -1) 4 > $O/$3 2>&1
rm -rf $O/selkt
head -45 $O/$3 | cut -c1-150
