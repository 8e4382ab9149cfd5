"""Print the headline fields of bench.py JSON lines (developer helper)."""
import json
import sys

for f in sys.argv[1:]:
    try:
        txt = open(f).read()
        line = [l for l in txt.splitlines() if l.startswith('{')][0]
        d = json.loads(line)
        print(f, '%.3f M/s' % (d['value'] / 1e6), '%.1f ms' % d['ms_per_step'],
              'roofline', round(d.get('roofline', {}).get('frac') or 0, 3),
              'scan', round(d.get('roofline_gae_scan', {}).get('frac') or 0, 3),
              'agg', round(d.get('mfma_aggregate', {}).get('frac') or 0, 3))
        for k in d.get('kernels', []):
            print('    %-86s %5d x %7.2f us = %7.2f ms' % (
                k['kernel'][:86], k['launches'], k['avg_us'], k['total_ms']))
    except Exception as e:  # noqa
        print(f, 'ERR', e, txt[-800:])
