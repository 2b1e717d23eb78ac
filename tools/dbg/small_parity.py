"""B > 2 x CUs at small N: the <2,1> instance against the oracle on a strided sample (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_fullbatch import run_full
for N in (1, 3, 8, 12, 15):
    run_full(1024, N, 3, [0, 1, 255, 256, 511, 512, 767, 1022, 1023])
    print("N", N, "ok")
