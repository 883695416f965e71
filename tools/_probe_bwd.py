import ctypes, os, subprocess, sys
import numpy as np
# run a few training iterations with the probe build, then read the stamps
sys.argv = ['bench_train.py', '2200', '5']
sys.path.insert(0, os.path.dirname(__file__))
exec(open(os.path.join(os.path.dirname(__file__), 'bench_train.py')).read())
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ['NRC_LIB_PATH'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', lib.nrc_debug_bwd_probe(buf))
a = np.array(buf[:], dtype=np.int64).reshape(2, 128)
for net, name in ((0, 'density'), (1, 'colour')):
    t0 = a[net, 0]
    print(name, 'entry->weights', a[net, 1] - t0)
    for t in range(8):
        pb = 10 + 10 * t
        row = a[net, pb:pb + 8]
        if row[0] == 0: break
        print(name, 'tile', t, 'start', row[0] - t0, 'phases', list(np.diff(row)))
    print(name, 'loop end', a[net, 120] - t0, 'flush', a[net, 121] - a[net, 120])
