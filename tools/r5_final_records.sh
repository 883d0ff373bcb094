#!/bin/bash
# Round-5 records in one GPU call: bench lines of the four configs (+ the 600-scan shape and the shard sizes), kernel
# trace of the default bench command, counter passes (HBM traffic; SQ issue counters; matrix-pipe counters) for the
# one-wave kernel of config 3 and for the split form at 600 scans; bench line and kernel trace of the four-wave form at 1 200 scans.
# (counter passes at 98 304 voxels = six whole rounds: every launch of the dominant kernel, partitioned call or alone, then has
# the same 6 144 waves -- a partitioned call of 100 000 sizes its grid for 6 250 of which 106 leave at once)
# Usage (on the GPU box, from the repo root):  bash tools/r5_final_records.sh <tag>
# Everything lands under gpurun_out/<tag>_*; copy what is to be kept into profiles/.
set -u
tag=${1:-r5k}
out=$PWD/gpurun_out
mkdir -p "$out"
for c in 2 4 5; do
  python3 bench.py --config $c > "$out/${tag}_bench_c$c.json" 2> "$out/${tag}_bench_c$c.err" || exit 1
  echo "config $c done"
done
python3 bench.py > "$out/${tag}_bench_c3.json" 2> "$out/${tag}_bench_c3.err" || exit 1
echo "config 3 done"
python3 bench.py --scans 600 --voxels 50000 --cpu-seconds 4 > "$out/${tag}_bench_600_scans.json" 2> "$out/${tag}_bench_600_scans.err" || exit 1
python3 bench.py --scans 1200 --voxels 16384 --cpu-seconds 4 > "$out/${tag}_bench_1200_scans.json" 2> "$out/${tag}_bench_1200_scans.err" || exit 1
echo "long series done"
for v in 12500 25000 50000; do
  python3 bench.py --voxels $v --cpu-seconds 0 --busy-seconds 0 > "$out/${tag}_bench_${v}_voxels.json" 2> "$out/${tag}_bench_${v}_voxels.err" || exit 1
done
python3 bench.py --config 4 --voxels 6250 --cpu-seconds 0 > "$out/${tag}_bench_c4_6250_voxels.json" 2> "$out/${tag}_bench_c4_6250_voxels.err" || exit 1
echo "shard sizes done"
run() {  # name, bench args (quoted), rocprofv3 args...
  local name=$1 bargs=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$out/${tag}_$name" -o run -- python3 "$out/../bench.py" $bargs --steps 3 --warmup 1 --cpu-seconds 0 --busy-seconds 0 > "$out/${tag}_$name.json" 2> "$out/${tag}_$name.err" ) || { echo "$name failed"; return 1; }
  echo "$name done"
}
run trace "" --kernel-trace --stats || exit 1
run fetch "--voxels 98304" --pmc FETCH_SIZE || exit 1
run write "--voxels 98304" --pmc WRITE_SIZE || exit 1
run sq "--voxels 98304" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY || exit 1
run mfma "--voxels 98304" --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS || exit 1
run trace600 "--scans 600 --voxels 49152" --kernel-trace --stats || exit 1
run trace1200 "--scans 1200 --voxels 16384" --kernel-trace --stats || exit 1
run sq600 "--scans 600 --voxels 49152" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY || exit 1
run mfma600 "--scans 600 --voxels 49152" --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS || exit 1
PB_PMC_KERNEL=fista_mfma_kernel python3 tools/summarise_pmc.py traffic "$out/${tag}_fetch" "$out/${tag}_write" "$out/${tag}_pmc_hbm_traffic.json" && \
PB_VOXELS_PER_WAVE=16 PB_PMC_KERNEL=fista_mfma_kernel python3 tools/summarise_pmc.py sq "$out/${tag}_sq" "$out/${tag}_pmc_sq.json" && \
PB_PMC_KERNEL=fista_mfma_kernel python3 tools/summarise_pmc.py sq "$out/${tag}_mfma" "$out/${tag}_pmc_mfma.json" && \
PB_VOXELS_PER_WAVE=8 PB_PMC_KERNEL=fista_mfma2_kernel python3 tools/summarise_pmc.py sq "$out/${tag}_sq600" "$out/${tag}_pmc_sq_600_scans.json" && \
PB_PMC_KERNEL=fista_mfma2_kernel python3 tools/summarise_pmc.py sq "$out/${tag}_mfma600" "$out/${tag}_pmc_mfma_600_scans.json"
find "$out/${tag}_trace" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
find "$out/${tag}_trace600" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats_600_scans.csv" \;
find "$out/${tag}_trace1200" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats_1200_scans.csv" \;
rm -rf "$out/${tag}_trace" "$out/${tag}_trace600" "$out/${tag}_trace1200" "$out/${tag}_fetch" "$out/${tag}_write" "$out/${tag}_sq" "$out/${tag}_mfma" "$out/${tag}_sq600" "$out/${tag}_mfma600"
echo "all done"
