#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root:   bash scripts/profile_round.sh <tag> [bench args]
#   kernel trace of bench.py (last 10 steps)                      -> gpurun_out/prof_<tag>/kernel_trace.txt
#   PMC passes (one counter group per run, no trace domains)      -> gpurun_out/prof_<tag>/pmc_<group>.txt
#   ray_knn FETCH_SIZE at P = 10k and 30k (scripts/bench_knn.py)   -> gpurun_out/prof_<tag>/pmc_knn_fetch.txt
# Copy what should be judged into profiles/.
TAG=${1:-r02}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
KEEP="ray_knn,mlp_chain,gemm_tn_h3,gemm_nt_h3,tail_,features_,segment_reduce,conv3x3,pairs_,upconv,maxpool,conv1x1,adam_,slab_reduce"
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 "$@" > $OUT/bench_under_trace.json 2> $OUT/kt.err
python3 scripts/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) last:10 > $OUT/kernel_trace.txt
pass() {   # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" -d $OUT/$name -o $name -- python3 bench.py --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 > /dev/null 2> $OUT/$name.err
    python3 scripts/rocpd_pmc.py $(find $OUT/$name -name "*.db") --keep $KEEP --top 30 > $OUT/pmc_$name.txt
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq_time SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
pass sq_insts SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU
pass l1 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TOTAL_CACHE_ACCESSES
pass grbm GRBM_GUI_ACTIVE
: > $OUT/pmc_knn_fetch.txt
for P in 10000 30000; do
    rocprofv3 --pmc FETCH_SIZE -d $OUT/knn$P -o knn$P -- python3 scripts/bench_knn.py $P > $OUT/knn_times_$P.txt 2> $OUT/knn$P.err
    echo "# P = $P points, R = 25,600 rays, k = 20 (scripts/bench_knn.py $P: 2 point orders x 13 launches)" >> $OUT/pmc_knn_fetch.txt
    python3 scripts/rocpd_pmc.py $(find $OUT/knn$P -name "*.db") --keep ray_knn --top 4 >> $OUT/pmc_knn_fetch.txt
    cat $OUT/knn_times_$P.txt >> $OUT/pmc_knn_fetch.txt
done
find $OUT -name "*.db" -delete          # (the databases are large; the text summaries are what travels back)
ls -la $OUT
