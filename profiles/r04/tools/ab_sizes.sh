#!/bin/bash
# A/B of build variants at larger genome sizes (fan-outs 2^17 / 2^18), inside one gpurun call:
#   bash tools/ab_build.sh NAME [-DFLAG ...]   (here, for every variant)
#   gpurun -- 'bash tools/ab_sizes.sh VARIANT_A VARIANT_B'
ROOT=$(pwd); LIB=$ROOT/krisp_amd/libkrisp_hip.so; cp $LIB /tmp/orig.so
OK=""
for v in "$@"; do
  cp $ROOT/krisp_amd/variants/$v.so $LIB
  if timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not wide" > $ROOT/gpurun_out/absz_$v.test.log 2>&1; then OK="$OK $v"
  else echo "$v FAILED tests"; tail -5 $ROOT/gpurun_out/absz_$v.test.log; fi
done
for r in 1 2; do for v in $OK; do cp $ROOT/krisp_amd/variants/$v.so $LIB
 for len in 100000000 200000000; do python3 bench.py --no-cpu-baseline --length $len --steps 5 --warmup 2 > $ROOT/gpurun_out/absz.json 2>/dev/null; python3 -c "
import json
d=json.loads(open('$ROOT/gpurun_out/absz.json').read().strip().splitlines()[-1]); st=d['roofline']['stage_ms_per_step_calibration']
print('$v', $len, 'ms/step %.3f G/s %.2f' % (d['ms_per_step'], d['value']/1e9), ' '.join(f'{k}={x:.2f}' for k, x in st.items() if x > 0.3))"; done; done; done
cp /tmp/orig.so $LIB
