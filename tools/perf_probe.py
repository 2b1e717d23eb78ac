"""Times the individual C-ABI calls (propagate / update_feat(M) / step) with HIP events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import vi_ekf_amd as v  # noqa: E402
from vi_ekf_amd import scene  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    dev = torch.device("cuda:0")
    sc = scene.make_scene(B, N, 4, seed=3)
    g = v.BatchVIEKF(B, N, sc["params"])
    from vi_ekf_amd import capi
    if kernel in (3, 5):      # the tile family whatever the batch size: 3 one filter per workgroup, 5 the paired form
        g.set_tuning(capi.TUNE_TILES, 2 if kernel == 3 else 3)
    elif kernel == 4:    # the on-chip families without the tile family
        g.set_tuning(capi.TUNE_TILES, 0)
    elif kernel:
        g.set_kernel(kernel)
    for kv in sys.argv[5:]:         # KEY=VALUE pairs for viekf_batch_set_tuning
        key, value = kv.split("=")
        g.set_tuning(int(key), int(value))
    print(g.describe())
    g.use_torch_stream()
    d = {k: torch.tensor(sc[k], device=dev) for k in ("u", "z", "dt", "slot", "R")}
    pix = torch.tensor(np.ascontiguousarray(sc["pix"].transpose(1, 0, 2)), device=dev)
    nan = torch.full((B,), float("nan"), dtype=torch.float64, device=dev)
    for i in range(N):
        g.init_feature(pix[i], nan)
    res = torch.empty((B, N), dtype=torch.int32, device=dev)

    def timeit(fn, name):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-28s %9.4f ms   (%.1f us per filter-round of 256 CUs)" % (name, ms, ms * 1e3 / max(1, (B + 255) // 256)))
        return ms

    timeit(lambda: g.propagate(d["u"][0], d["dt"]), "propagate")
    for M in (1, 2, 6, 10, 16, 25, N):
        timeit(lambda: g.update_feat(d["z"][0][:, :M].contiguous(), d["slot"][:, :M].contiguous(), d["R"],
                                     result=res[:, :M].contiguous()), "update_feat M=%d" % M)
    timeit(lambda: g.step(d["u"][0], d["dt"], d["z"][0], d["slot"], d["R"], result=res), "step")


if __name__ == "__main__":
    main()
