#!/bin/bash
for f in "$@"; do echo "$f: $(python -c "
import json,sys
try:
    l=[x for x in open('$f') if x.startswith('{')][0]; d=json.loads(l)
    r=d.get('roofline'); 
    print(round(d['value']/1e6,3),'M obs/s', round(d['ms_per_step'],4),'ms', ('raster %.1f us step %.1f sort %.1f' % (r['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3, d['kernels']['frame_setup_and_sort']['avg_launch_ms']*1e3)) if r else '')
except Exception as e: print('ERR',e, open('$f').read()[-400:])
")"; done
