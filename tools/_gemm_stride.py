import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lmrs_amd as L
rng = np.random.default_rng(0)
for n, o in [(8192, 2048), (8448, 2048), (2048, 2048), (2304, 2048), (2048, 16384), (2304, 16384)]:
    sl = 512
    wq = rng.integers(-127, 128, (o, n), dtype=np.int8); ws = rng.random(o * n // 128, dtype=np.float32)
    xq = rng.integers(-127, 128, (sl, n), dtype=np.int8); xs = rng.random(sl * n // 128, dtype=np.float32)
    for _ in range(2):
        out = L.matmul_q8(xq.reshape(-1), xs, wq.reshape(-1), ws, n, o, sl=sl)
    print(n, o, float(np.abs(out).sum()))
