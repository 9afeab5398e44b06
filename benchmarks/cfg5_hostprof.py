"""Host-side profile of the cfg5 sweep (cProfile over the timed sweeps of benchmarks/secondary.py:cfg5)."""
import argparse
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from benchmarks import secondary

    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=None)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--no-cpu", action="store_true", default=True)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    pr = cProfile.Profile()
    pr.enable()
    out = secondary.cfg5(args, torch)
    pr.disable()
    print(out["ms_per_step"], "ms per step under the profiler", file=sys.stderr)
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(args.top)


if __name__ == "__main__":
    main()
