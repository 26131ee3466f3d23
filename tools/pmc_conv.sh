#!/bin/bash
# SQ counters of single conv shapes per tile (GPU box): tools/pmc_conv.sh OUTDIR "10 17 19"
# one rocprofv3 --pmc pass per (tile, shape); folded by tools/pmc_sq_summary.py
out=$1; tiles=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in $tiles; do
  for shape in "--h 256 --c1 128 --cout 128 --temb" "--h 128 --c1 256 --cout 256 --temb"; do
    tag=$(echo $shape | tr -d ' -')
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
      --output-format csv -d $out/t${t}_$tag -- python3 tools/conv_one.py $shape --tile $t --stats --reps 4 > $out/t${t}_$tag.log 2>&1 || exit 1
    python3 tools/pmc_sq_summary.py $out/t${t}_$tag $out/t${t}_$tag.json conv
  done
done
