#!/usr/bin/env python3
"""tools/probe_cycles.py mlp_bwd|gb_accumulate -- shader-clock stamps of ONE wave at the phases of a kernel (developer aid).

Needs the probe build of the library: `bash tools/build_variant.sh probe ngp_net.hip -DNRC_BWD_PROBE`, then
`NRC_LIB_PATH=_ab/probe.so python tools/probe_cycles.py mlp_bwd`.  The build adds `s_memtime` stamps (wave 0 of one workgroup) to
k_nwie_bwd (with an `s_waitcnt 0` in front of each, so the phases are separated but the prefetch no longer overlaps) and to
k_gb_accumulate (no waits: the real timeline), and an extra entry point that copies the stamps out.  This is how round 3 found that
k_nwie_bwd spends 3.4 of 6.1 K cycles per tile in the LDS transposes of its weight-gradient products, and that k_gb_accumulate is
bound by the LDS atomic rate (44 K of 70 K cycles) with a 13 K prologue and a 12 K read-modify-write flush."""
import ctypes, os, sys
import numpy as np

mode = sys.argv[1] if len(sys.argv) > 1 else 'mlp_bwd'
sys.argv = ['bench_train.py', '2200', '3']
exec(open(os.path.join(os.path.dirname(__file__), 'bench_train.py')).read())
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ['NRC_LIB_PATH'])
buf = (ctypes.c_ulonglong * 256)()
rc = lib.nrc_debug_bwd_probe(buf)
assert rc == 0, rc
a = np.array(buf[:], dtype=np.int64).reshape(2, 128)
if mode == 'mlp_bwd':
    for net, name in ((0, 'density'), (1, 'colour')):
        t0 = a[net, 0]
        print(name, 'entry -> weight fragments', a[net, 1] - t0)
        for t in range(8):
            row = a[net, 10 + 10 * t:18 + 10 * t]
            if row[0] == 0:
                break
            print(name, 'tile', t, 'starts at', row[0] - t0, 'phases (dZ out, dW out, dH, dW1, dH0, dW0, d_in + stores):', [int(v) for v in np.diff(row)])
        print(name, 'loop ends at', a[net, 120] - t0, 'weight-gradient flush', a[net, 121] - a[net, 120])
else:
    t0 = a[0, 0]
    names = {1: 'first records requested', 2: 'slice cleared', 3: 'scale known', 10: 'batch 1 requested', 11: 'batch 0 added', 12: 'batch 2 requested',
             13: 'batch 1 added', 40: 'all waves done', 41: 'slice written back'}
    for k, n in names.items():
        if a[0, k]:
            print(f'{n:26s} {a[0, k] - t0:8d} cycles after entry')
