#!/bin/bash
# usage: bench_short.sh <lib or ""> <workload>: prints value, compress in-loop, decompress, isolated
L=$1; W=$2
if [ -n "$L" ]; then export WAH_LIB_PATH=$L; fi
python bench.py --no-cpu-baseline --steps 20 --warmup 3 --workload $W | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', '$W', 'value', d['value'], 'comp', d['compress_ms']['avg'], 'dec', d['decompress_ms']['avg'], 'iso', d['roofline']['launch_ms_isolated'], 'frac', d['roofline']['frac'])"
