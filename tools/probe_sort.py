#!/usr/bin/env python3
"""tools/probe_sort.py [N] -- shader-clock stamps of thread 0 of three tiles (first, #100, last) at the phases of the last depth-sort pass.
Needs the probe build: `bash tools/build_variant.sh sortprobe gs_raster.hip -DNRC_SORT_PROBE`, then `NRC_LIB_PATH=_ab/sortprobe.so python tools/probe_sort.py`."""
import ctypes, os, sys
import numpy as np
sys.argv = ['gs_fwd_only.py'] + (sys.argv[1:2] or ['1000000']) + ['5']
exec(open(os.path.join(os.path.dirname(__file__), 'gs_fwd_only.py')).read())
lib = ctypes.CDLL(os.environ['NRC_LIB_PATH'])
buf = (ctypes.c_ulonglong * 48)()
assert lib.nrc_debug_sort_probe(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(3, 16)
names = ['entry', 'loads issued', 'loads landed', 'ranked', 'barrier', 'published + scans', 'keys staged', 'group words checked', 'in-group sum', 'group totals', 'barrier', 'stores issued', 'stores drained']
t_first = a[:, 0].min()
for b, label in enumerate(('tile 0', 'tile 100', 'last tile')):
    print(label, 'entered', a[b, 0] - t_first, 'cycles after the first of the three;', ' | '.join(f'{names[k]} +{a[b, k] - a[b, k - 1]}' for k in range(1, 13)), '| total', a[b, 12] - a[b, 0])
