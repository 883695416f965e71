import ctypes, os, sys
import numpy as np
sys.argv = ['bench_train.py', '2200', '3']
sys.path.insert(0, os.path.dirname(__file__))
exec(open(os.path.join(os.path.dirname(__file__), 'bench_train.py')).read())
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ['NRC_LIB_PATH'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', lib.nrc_debug_bwd_probe(buf))
a = np.array(buf[:], dtype=np.int64).reshape(2, 128)[0]
t0 = a[0]
for k in (1, 2, 3, 10, 11, 12, 13, 14, 15, 40, 41):
    if a[k]: print(k, a[k] - t0)
