# dev tool (GPU box): SQ counters of the halo wgrad kernel on one shape.  usage: bash tools/pmc_wgrad.sh [dbg] [pipe values] [shape]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DBG=${1:-0}; VAL=${2:-1}; SH=${3:-512-512-32}
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INSTS_VALU"; do
  d=$R/gpurun_out/pmc_wg/$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python $R/tools/wgrad_pipe_ab.py --only $SH --values $VAL --rounds 1 --reps 2 --dbg $DBG > /dev/null 2>&1
  f=$(find $d -name "*.db" | head -1)
  echo "== $set"; python $R/tools/rocpd_counters.py $f wgrad3x3 2>&1 | tail -5
done
