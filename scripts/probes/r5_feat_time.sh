# features_fwd / _bwd kernel time in the bench step for variant libraries: LIBS="base tag ..."
mkdir -p gpurun_out/r5b; export TMPDIR=/tmp
for L in ${LIBS}; do
  if [ "$L" = base ]; then unset PAPR_HIP_LIB; else export PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$L.so; fi
  O=gpurun_out/r5b/ft_$L; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace -d $O/kt -o kt -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 > /dev/null 2> $O/err
  python3 scripts/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) last:5 > $O/kernel_trace.txt; find $O -name "*.db" -delete
  echo "lib $L: $(grep "features_fwd\|features_bwd\|segment_reduce\|tail_\|conv3x3" $O/kernel_trace.txt | awk '{print $1, $4}' | tr '\n' ' ')"
done
