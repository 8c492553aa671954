#!/bin/bash
# Where does the implicit GEMM's K loop lose its time?  Three separate rocprofv3 --pmc passes (SQ issue / wait split and MFMA
# busy / co-execution; SQ per-instruction-class activity and queue levels; TA / TCP / TCC) over the per-layer conv bench,
# summarised per kernel (tools/sq_summarize.py).  Usage: gpu_loop_counters.sh TAG [config]
TAG="${1:-loopctr}"; CFG="${2:-moving-gif}"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python $PWD/tools/conv_bench.py --config $CFG --batch 32 --iters 3"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
P2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE"
P4="GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i + 1))
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OLDPWD/$OUT/p$i" -o c -- $CMD > "$OLDPWD/$OUT/p$i.log" 2>&1 ); echo "pass $i rc=$?"
  python tools/sq_summarize.py "$OUT/p$i" > "$OUT/pass$i.txt" 2>&1
  head -12 "$OUT/pass$i.txt" | cut -c1-400
  find "$OUT" -name "*counter_collection*" -size +6M -delete; find "$OUT" -name "*kernel_trace*" -size +4M -delete
done
