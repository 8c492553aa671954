#!/bin/bash
# SQ counter pass over the per-layer conv benchmark (taichi by default)
TAG="${1:-sq}"; CFG="${2:-taichi}"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python $PWD/tools/conv_bench.py --config $CFG --batch 32 --iters 3"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OLDPWD/$OUT/sq" -o s -- $CMD > "$OLDPWD/$OUT/sq.log" 2>&1 ); echo "rc=$?" | tee "$OUT/summary.txt"
python tools/sq_summarize.py "$OUT/sq" 2>&1 | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d "$OLDPWD/$OUT/sq2" -o s -- $CMD > "$OLDPWD/$OUT/sq2.log" 2>&1 ); echo "rc=$?" | tee -a "$OUT/summary.txt"
python tools/sq_summarize.py "$OUT/sq2" 2>&1 | tee -a "$OUT/summary.txt"
tail -5 "$OUT/sq.log" | cut -c1-200 >> "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +8M -delete
