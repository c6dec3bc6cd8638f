for i in 1 2 3; do for v in 0 1; do
  if [ $v = 1 ]; then export C8P_BENCH_NOPROFILE=1; else unset C8P_BENCH_NOPROFILE; fi
  python bench.py --steps 20 --warmup 5 --no-cpu --legs none --detail /tmp/abp.json > /dev/null 2>&1; python -c "
import json; j=json.load(open('/tmp/abp.json')); print('noprofile=$v steps20', round(j['value']), round(j['ms_per_step'],4))"
  python bench.py --steps 200 --no-cpu --legs none --detail /tmp/abp.json > /dev/null 2>&1; python -c "
import json; j=json.load(open('/tmp/abp.json')); print('noprofile=$v steps200', round(j['value']), round(j['ms_per_step'],4))"
done; done
