#!/bin/bash
# Round evidence on the GPU box (run through gpurun from the repo root), in STAGES so that one call stays short:
#   bash tools/collect_evidence.sh r03 tests|bench|prof|cutprof|cutpmc|lines
# Outputs -> gpurun_out/<tag>_*; copy what should be judged into profiles/.  rocprofv3: kernel-trace only next to --pmc, one counter set
# per pass.  Every profiled command carries --no-cut-leg / a bounded step count: a PMC pass serialises every dispatch.
TAG=${1:-r03}
STAGE=${2:-bench}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
BUILD="csrc sha1 $(cat joligen_amd/csrc/*.hip joligen_amd/csrc/*.h | sha1sum | cut -c1-12)"   # .git does not travel to the box
echo "build $BUILD stage $STAGE" >> $O/${TAG}_evidence.log
db() { ls $O/${TAG}_$1/*.db 2>/dev/null | head -1; }
meta() {  # file, command text
python - <<PY
import json
try:
    d=json.load(open("$1")); d["_meta"]={"build":"$BUILD","command":"$2"}
    json.dump(d,open("$1","w"),indent=1)
except Exception as e: print("pmc meta", e)
PY
}
case $STAGE in
tests)
  timeout 600 python -m pytest tests -v -m gpu -p no:cacheprovider > $O/${TAG}_gpu_pytest_full.log 2>&1
  tail -3 $O/${TAG}_gpu_pytest_full.log | tee -a $O/${TAG}_evidence.log ;;
bench)
  timeout 500 python bench.py 2> $O/${TAG}_bench.err | tail -1 > $O/${TAG}_bench_line.json
  cut -c1-600 $O/${TAG}_bench_line.json ;;
prof)
  timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cut-leg --dump-kernel-timing $O/${TAG}_per_layer_kernel_timing.txt > /dev/null 2>&1
  export JG_TRACE_MARK=1      # steady-state window for tools/rocpd_stats.py (4 timed steps)
  B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-cut-leg"
  cd /tmp
  timeout 120 rocprofv3 --kernel-trace -d $O/${TAG}_kt -o kt -- $B > $O/${TAG}_kt.log 2>&1
  JG_WGRAD_STREAM=0 timeout 120 rocprofv3 --kernel-trace -d $O/${TAG}_ki -o ki -- $B > $O/${TAG}_ki.log 2>&1
  timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pf -o pf -- $B > $O/${TAG}_pf.log 2>&1
  timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pw -o pw -- $B > $O/${TAG}_pw.log 2>&1
  timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/${TAG}_pb -o pb -- $B > $O/${TAG}_pb.log 2>&1
  cd $R
  python tools/rocpd_stats.py $(db kt) 4 > $O/${TAG}_kernel_stats.md 2>> $O/${TAG}_evidence.log
  python tools/rocpd_stats.py $(db ki) 4 > $O/${TAG}_kernel_stats_isolated.md 2>> $O/${TAG}_evidence.log
  python tools/rocpd_overlap.py $(db kt) > $O/${TAG}_overlap.txt 2>> $O/${TAG}_evidence.log
  python tools/rocpd_pmc.py $(db pf) $(db pw) $O/${TAG}_pmc.json > $O/${TAG}_pmc_hbm_traffic.md 2>> $O/${TAG}_evidence.log
  python tools/rocpd_busy.py $(db pb) > $O/${TAG}_mfma_busy.md 2>> $O/${TAG}_evidence.log
  meta $O/${TAG}_pmc.json "bench.py --steps 3 --warmup 1 --no-cut-leg (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes)"
  rm -rf $O/${TAG}_kt $O/${TAG}_ki $O/${TAG}_pf $O/${TAG}_pw $O/${TAG}_pb $O/${TAG}_*.log.bak
  head -12 $O/${TAG}_kernel_stats.md; head -8 $O/${TAG}_mfma_busy.md ;;
cutprof)
  export JG_TRACE_MARK=1
  CB="python $R/bench.py --model cut --netG segformer_attn_conv --netDs projected_d,basic --proj vitsmall --batch 16 --steps 4 --warmup 5 --no-cpu-baseline --no-kernel-timing"
  timeout 120 python bench.py --model cut --netG segformer_attn_conv --netDs projected_d,basic --proj vitsmall --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --dump-kernel-timing $O/${TAG}_cut_per_layer_kernel_timing.txt > /dev/null 2>&1
  cd /tmp
  timeout 150 rocprofv3 --kernel-trace -d $O/${TAG}_ckt -o kt -- $CB > $O/${TAG}_ckt.log 2>&1
  cd $R
  python tools/rocpd_stats.py $(db ckt) 4 > $O/${TAG}_cut_kernel_stats.md 2>> $O/${TAG}_evidence.log
  python tools/rocpd_overlap.py $(db ckt) > $O/${TAG}_cut_overlap.txt 2>> $O/${TAG}_evidence.log
  python tools/rocpd_stats.py $(db ckt) 4 --by-queue > $O/${TAG}_cut_kernel_stats_by_queue.md 2>> $O/${TAG}_evidence.log
  rm -rf $O/${TAG}_ckt
  head -14 $O/${TAG}_cut_kernel_stats.md ;;
cutpmc)
  # HBM traffic of the CUT step's dominant instances only (--kernel-include-regex): a PMC pass serialises every counted dispatch, the
  # whole 4000-launch step did not finish in 40 minutes in round 3
  CB="python $R/bench.py --model cut --netG segformer_attn_conv --netDs projected_d,basic --proj vitsmall --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
  RX="wgrad_tn_tr_kernel|conv_nt_glds_kernel"
  cd /tmp
  timeout 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$RX" -d $O/${TAG}_cpf -o pf -- $CB > $O/${TAG}_cpf.log 2>&1
  timeout 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$RX" -d $O/${TAG}_cpw -o pw -- $CB > $O/${TAG}_cpw.log 2>&1
  cd $R
  python tools/rocpd_pmc.py $(db cpf) $(db cpw) $O/${TAG}_cut_pmc.json > $O/${TAG}_cut_pmc_hbm_traffic.md 2>> $O/${TAG}_evidence.log
  meta $O/${TAG}_cut_pmc.json "bench.py --model cut --netG segformer_attn_conv --netDs projected_d,basic --proj vitsmall --batch 16 --steps 2 --warmup 1 (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, --kernel-include-regex '$RX')"
  rm -rf $O/${TAG}_cpf $O/${TAG}_cpw
  head -12 $O/${TAG}_cut_pmc_hbm_traffic.md ;;
lines)
  for cfg in "resnet basic" "segformer_attn_conv basic" "segformer_attn_conv projected_d,basic" "mobile_resnet_attn projected_d,basic"; do
    set -- $cfg
    timeout 120 python bench.py --model cut --netG $1 --netDs $2 --batch 16 --steps 10 --warmup 3 --no-kernel-timing --no-cpu-baseline 2>/dev/null | tail -1 >> $O/${TAG}_cut_bench_lines.json
  done
  # BASELINE configs[3] (512x512, non-efficient, batch 8) and configs[4] (cm_model, batch 64)
  timeout 120 python bench.py --size 512 --batch 8 --efficient 0 --steps 20 --no-cut-leg --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_c4_512_bench_line.json
  timeout 120 python bench.py --model cm --batch 64 --steps 20 --no-cut-leg --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_cm_bench_line.json
  cut -c1-200 $O/${TAG}_cut_bench_lines.json $O/${TAG}_c4_512_bench_line.json $O/${TAG}_cm_bench_line.json ;;
esac
