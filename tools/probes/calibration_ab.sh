for f in "" "--no-calibration" "" "--no-calibration"; do
python bench.py --no-extras --no-cpu-baseline $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
k=[v['us_per_launch'] for n,v in r['kernels'].items() if 'f32in' in n][0]
print('$f', d['ms_per_step'], 'first layer us', k, 'all_conv', r['all_conv_frac'], 'fc1 frac', r['frac'], d['device_calibration'] and (d['device_calibration']['copy_TBps'], d['device_calibration']['mfma_bf16_TFLOPs']))
"
done
