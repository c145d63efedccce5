ROOT=$(pwd); LIB=$ROOT/krisp_amd/libkrisp_hip.so; cp $LIB /tmp/orig.so
for v in p2small p2big; do cp $ROOT/krisp_amd/variants/$v.so $LIB; timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not wide" > $ROOT/gpurun_out/abc4_$v.test.log 2>&1 || { echo "$v FAILED tests"; tail -5 $ROOT/gpurun_out/abc4_$v.test.log; }; done
for r in 1 2; do for v in p2small p2big; do cp $ROOT/krisp_amd/variants/$v.so $LIB
 for len in 100000000 200000000; do python3 bench.py --no-cpu-baseline --length $len --steps 5 --warmup 2 > $ROOT/gpurun_out/abc4.json 2>/dev/null; python3 -c "
import json
d=json.loads(open('$ROOT/gpurun_out/abc4.json').read().strip().splitlines()[-1]); st=d['roofline']['stage_ms_per_step_calibration']
print('$v', $len, 'ms/step %.3f G/s %.2f' % (d['ms_per_step'], d['value']/1e9), 'hist2=%.2f scatter2=%.2f localsort=%.2f' % (st['hist2'], st['scatter2'], st['localsort']))"; done; done; done
cp /tmp/orig.so $LIB
