#!/bin/bash
# SQ counter passes for the four-wave form at 1 200 scans (as tools/r5_final_records.sh does for the other two kernels).
set -u
tag=${1:-r5m}
out=$PWD/gpurun_out
mkdir -p "$out"
run() {
  local name=$1 bargs=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$out/${tag}_$name" -o run -- python3 "$out/../bench.py" $bargs --steps 3 --warmup 1 --cpu-seconds 0 --busy-seconds 0 > "$out/${tag}_$name.json" 2> "$out/${tag}_$name.err" ) || { echo "$name failed"; return 1; }
  echo "$name done"
}
run sq1200 "--scans 1200 --voxels 16384" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY || exit 1
run mfma1200 "--scans 1200 --voxels 16384" --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS || exit 1
run lds1200 "--scans 1200 --voxels 16384" --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY || exit 1
PB_VOXELS_PER_WAVE=4 PB_PMC_KERNEL=fista_mfma4_kernel python3 tools/summarise_pmc.py sq "$out/${tag}_sq1200" "$out/${tag}_pmc_sq_1200_scans.json" && \
PB_PMC_KERNEL=fista_mfma4_kernel python3 tools/summarise_pmc.py sq "$out/${tag}_mfma1200" "$out/${tag}_pmc_mfma_1200_scans.json" && \
PB_PMC_KERNEL=fista_mfma4_kernel python3 tools/summarise_pmc.py sq "$out/${tag}_lds1200" "$out/${tag}_pmc_lds_1200_scans.json"
rm -rf "$out/${tag}_sq1200" "$out/${tag}_mfma1200" "$out/${tag}_lds1200"
echo "all done"
