#!/bin/bash
# Round 6 sweep: bench lines of cfg3 / cfg5 / cfg2 under environment variants ("NAME:VAR=V,VAR=V ..."), after the parity tests of the touched kernels.
set -u
OUT=gpurun_out/${1:-r06f}
shift
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -5 > $OUT/tests.txt
timeout 900 python -m pytest tests/test_full_model_gpu.py -x -q -k "kitti or cityscapes or cfg3 or cfg5 or cfg1 or cvppp" 2>&1 | tail -5 >> $OUT/tests.txt
cat $OUT/tests.txt
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  for c in ${CFGS:-cfg3 cfg5 cfg2}; do
    extra="--config $c"; [ $c = cfg2 ] && extra="--no-train-object"
    env $(echo $envs | tr ',' ' ') timeout 600 python bench.py $extra --no-cpu-baseline > $OUT/bench_${c}_$name.json 2>> $OUT/bench.err
  done
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(f,'ERR'); continue
    r=j['roofline']
    print('%-40s value %7.0f enc %6.1f (%6.1f) tail %6.1f (%s) lone %s' % (os.path.basename(f)[6:-5], j['value'], r['avg_us_per_launch_group'], r['as_launched']['avg_us_per_launch_group'], j['tail_us'], j.get('tail_us_as_launched'), j['config'].get('lone_batch_ms')))
    print('      ', ' '.join('%s:%s %.1f' % ('+'.join(map(str,l['layers'])), l.get('kernel',''), l['avg_us']) for l in r['layers']))
PY
