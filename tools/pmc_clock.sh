# dev tool (GPU box): effective shader clock under the MFMA kernels = GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md, DVFS give-back)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name, command...
  d=$R/gpurun_out/pmc_clk/$1; shift; rm -rf $d
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $d -o p -- "$@" > /dev/null 2>&1
  python $R/tools/rocpd_counters.py $(find $d -name "*.db" | head -1) 3x3 2>&1 | grep -v "^$"
}
echo "== wgrad 512-512-32 full"; run wg0 python $R/tools/wgrad_pipe_ab.py --only 512-512-32 --values 1 --rounds 1 --reps 3
echo "== wgrad 512-512-32 loop only (dbg 13)"; run wg13 python $R/tools/wgrad_pipe_ab.py --only 512-512-32 --values 1 --rounds 1 --reps 3 --dbg 13
echo "== fwd 512-512-32"; run fw python $R/tools/halo_pipe_ab.py --only 512-512-32 --values 1 --rounds 1 --reps 3
