set -x
mkdir -p gpurun_out/r6a
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r6a/driver_detail.json > gpurun_out/r6a/driver_line.json 2> gpurun_out/r6a/driver.err ) 2> gpurun_out/r6a/driver.time
wc -c gpurun_out/r6a/driver_line.json; cat gpurun_out/r6a/driver_line.json; cat gpurun_out/r6a/driver.time
python -c "
import json; d=json.load(open('gpurun_out/r6a/driver_detail.json')); print(d.get('leg_wall_s'), d.get('legs_skipped'), d.get('startup_s'), d.get('wall_s')); print([k for k in d if k.endswith('_error')], [ (k,d[k]) for k in d if k.endswith('_error')])"
