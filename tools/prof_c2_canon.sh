# rocprofv3 kernel statistics of `bench.py --config 2 --lanes 1` per library variant / switch (round 6: the canonical flank
# dictionary, positions per thread of k_hist8w):  bash tools/prof_c2_canon.sh name[:ENV=V] ...   (name = a file of krisp_amd/variants or "product")
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_c2prof; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for spec in "$@"; do
  v=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  unset KRISP_HIP_LIB KR_WIDE_CANON
  [ "$v" != product ] && [ -f $ROOT/krisp_amd/variants/$v.so ] && export KRISP_HIP_LIB=$ROOT/krisp_amd/variants/$v.so
  [ -n "$envs" ] && export $envs
  tag=$(echo $spec | tr ':=' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --config 2 --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > $OUT/$tag.log 2>&1
  find $OUT/$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/$tag
  grep k_hist8w $OUT/${tag}_kernel_stats.csv | head -2
done
