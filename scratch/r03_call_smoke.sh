#!/bin/bash
set -e
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_hbm']['frac'], d['roofline_gemm']['frac'], d['cpu_baseline']['value'], d['roofline_alt']['value_alt'])"
