set -e
for a in 1 2; do
  EXTRA_HIPCC_FLAGS="-DSYG_COLS_ABL=$a" ./build_lib.sh > /dev/null 2>&1
  export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl$a -- python3 tools/conv_prof.py conv > gpurun_out/abl$a.log 2>&1
done
echo done
