#!/bin/bash
# three bench workloads, compress / decompress / value per line
for wl in sparse clustered dense; do
  python bench.py --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', 'value', d['value'], 'compress', d['compress_ms']['avg'], 'decompress', d['decompress_ms']['avg'], 'indexed', d['roofline_decompress_indexed']['launch_ms'])"
done
