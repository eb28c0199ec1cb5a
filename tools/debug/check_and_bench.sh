#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_size.py tests/test_gpu_train.py -x -q 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-all 2>&1 | grep -v "^{" | head -24
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], 'E', d['config']['total_energy_rank0'])"
