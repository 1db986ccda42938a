# FETCH_SIZE / WRITE_SIZE of the AMP step's kernels (bench.py --amp), separate passes
OUT=gpurun_out/r5b/pmc_amp; mkdir -p $OUT
export TMPDIR=/tmp
KEEP="mlp_chain,gemm_tn_h3,tail_,features_,conv3x3"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/$C -o $C -- python3 bench.py --amp --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 > /dev/null 2> $OUT/$C.err
  python3 scripts/rocpd_pmc.py $(find $OUT/$C -name "*.db") --keep $KEEP --top 14 > $OUT/pmc_$C.txt
done
find $OUT -name "*.db" -delete
cat $OUT/pmc_FETCH_SIZE.txt | cut -c1-170
