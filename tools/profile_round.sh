#!/bin/bash
# The three rocprofv3 collections behind profiles/<tag>_* (GPU box).  usage: tools/profile_round.sh <tag>
# One program directly behind `--`, counters in passes of their own (never combined with a trace).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-line"
P="python3 bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-fp32-line"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $P > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $P > $out/pmc_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $out/pmc_sq -- $P > $out/pmc_sq.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/${tag}_pmc_hbm_traffic.json > $out/pmc_summary.log 2>&1
python3 tools/pmc_sq_summary.py $out/pmc_sq $out/${tag}_pmc_sq.json conv attn gn_ ddim_step > $out/pmc_sq_summary.log 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
tail -1 $out/trace.log | cut -c1-300
head -12 $out/${tag}_bench_kernel_stats.csv | cut -c1-160
