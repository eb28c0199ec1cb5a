for m in 0 32 16; do
python bench_train.py --no-cpu-baseline --steps 6 --warmup 2 --micro $m 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('micro=$m', round(d['value']), round(d['ms_per_step'],2), d['config']['workspace_gb'], d['config']['micro_batches'])"
done
