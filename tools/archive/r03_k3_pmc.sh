#!/bin/bash
# Round 3: counter evidence for K3 (k_matvec<LAP>) on the C5 problem -- which resource binds the u[col] gather.
# One rocprofv3 --pmc pass per counter group (separate runs, kernel trace only); a failing group (unknown counter, too many for the
# block's slots) does not stop the others.  Outputs: gpurun_out/r03_k3_pmc/<group>/ + a summary text per group.
#   usage (on the GPU box, from the repo root):  bash tools/archive/r03_k3_pmc.sh [probe.py]
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
PROBE=${1:-tools/archive/r03_k3_probe.py}
OUT=${K3_PMC_OUT:-$ROOT/gpurun_out/r03_k3_pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
declare -A PMCG
PMCG[tcp_req]="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
PMCG[tcp_stall]="TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
PMCG[tcp_fifo]="TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
PMCG[tcp_tlb]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum"
PMCG[tcc_hit]="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
PMCG[tcc_ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
PMCG[ta]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
PMCG[td]="TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum"
PMCG[sq_a]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
PMCG[sq_b]="SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES"
for g in ${K3_PMC_GROUPS:-tcp_req tcp_stall tcp_fifo tcp_tlb tcc_hit tcc_ea ta td sq_a sq_b}; do
  d=$OUT/$g
  rm -rf "$d"; mkdir -p "$d"
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc ${PMCG[$g]} -d "$d" -o pmc -- python "$ROOT/$PROBE" > "$d/run.log" 2>&1 )
  echo "group $g: exit $?" | tee -a "$OUT/status.txt"
  f=$(find "$d" -name '*.db' | head -1)
  if [ -n "$f" ]; then python "$ROOT/profiles/summarize_pmc_db.py" "$f" k_matvec k_mv > "$OUT/$g.txt" 2>&1; cat "$OUT/$g.txt"; find "$d" -name '*.db' -size +30M -delete; fi
done
