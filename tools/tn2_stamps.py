"""Diagnostic (not part of the product or the tests): s_memtime stamps of one workgroup of tn_topics_w2_kernel, per 64-row chunk.
Needs a library built with -DGDRF_TN2_STAMPS.  Slots: 0 chunk start, 1 DMA issued, 2 step 0 starts (fragment loads), 3 step 0 stage 1
starts, 4 step 1 starts, 5 step 1 stage 1 starts, 6 last MFMA issued, 7 behind the barrier."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
OUT = "gpurun_out/tn2_stamps.bin"
os.makedirs("gpurun_out", exist_ok=True)


def main():
    import tools.nt_trace as nt
    step = nt.build_step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    os.environ["GDRF_TN2_STAMP_FILE"] = OUT
    step()
    torch.cuda.synchronize()
    del os.environ["GDRF_TN2_STAMP_FILE"]
    a = np.fromfile(OUT, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    a = a[a[:, 0] > 0]
    a = a[4:-2]
    d = np.diff(a, axis=1)
    names = ["dma issue", "x reads + first splits", "step 0: frags + stage 0", "step 0: stages 1-9", "step 1: frags + stage 0", "step 1: stages 1-9", "wait + barrier"]
    print("chunks", len(a), " cycles per chunk (start to start):", np.median(np.diff(a[:, 0])))
    for i, n in enumerate(names):
        print(f"  {n:28s} median {np.median(d[:, i]):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
    print("  next chunk start - barrier:", np.median(a[1:, 0] - a[:-1, 7]))


if __name__ == "__main__":
    main()
