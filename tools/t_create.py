import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krisp_amd import _native as N
x = np.frombuffer(b"ACGT" * 12_500_000, dtype=np.uint8)
for rep in range(3):
    t = time.time(); e = N.Engine(); t1 = time.time() - t
    t = time.time(); e.set_params(25, 1, 2, max_bases=50_000_000); t2 = time.time() - t
    t = time.time(); e.upload(0, x); t3 = time.time() - t
    t = time.time(); e.sort(0); n = e.count(0); t4 = time.time() - t
    t = time.time(); e.close(); t5 = time.time() - t
    print(f"rep {rep}: create {t1:.4f} set_params {t2:.4f} upload {t3:.4f} sort+count {t4:.4f} close {t5:.4f}")
