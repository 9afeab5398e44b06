#!/usr/bin/env python3
"""Timeline of one workgroup from the phase stamps of `bench.py --stamps` (OMC_STAMPS_DUMP=file.npy): for every wave the
time of each stamp relative to the workgroup's first stamp, median over chains -- shows which wave the others wait for
at each barrier (relative shares per wave hide that: every wave's clock starts at its own entry).

    OMC_STAMPS_DUMP=gpurun_out/st.npy python3 bench.py --stamps --no-cpu --secondary-ms 0 ; python3 benchmarks/stamp_timeline.py gpurun_out/st.npy
"""
import sys

import numpy as np

NAMES = ["entry", "gamma draws", "fill b", "fill a", "moebius local", "moebius scan", "newton", "l", "fill rhs",
         "fwd map+scan", "fwd pass+rng", "bwd map+scan", "bwd pass", "tile+quad", "quad sum", "epilogue+store"]


def main():
    st = np.load(sys.argv[1])  # [C][16][16]
    nw = int((st[0, :, 0] > 0).sum())
    st = st[:, :nw, :]
    t0 = st[:, :, 0].min(axis=1)[:, None, None]
    rel = np.median(st - t0, axis=0)  # [wave][stamp]
    print("ticks since the workgroup's first wave entered, median over %d chains; columns: waves 0, 1, %d, %d, then min / max over waves"
          % (st.shape[0], nw // 2, nw - 1))
    for k in range(16):
        r = rel[:, k]
        print("  after %-16s w0 %7.0f  w1 %7.0f  w%-2d %7.0f  w%-2d %7.0f   min %7.0f (w%d)  max %7.0f (w%d)" % (
            NAMES[k], r[0], r[1], nw // 2, r[nw // 2], nw - 1, r[nw - 1], r.min(), int(r.argmin()), r.max(), int(r.argmax())))
    d = np.diff(rel, axis=1)
    print("phase length per wave (ticks): min / median / max over waves, slowest wave")
    for k in range(15):
        print("  %-16s %7.0f %7.0f %7.0f  w%d" % (NAMES[k + 1], d[:, k].min(), np.median(d[:, k]), d[:, k].max(), int(d[:, k].argmax())))


if __name__ == "__main__":
    main()
