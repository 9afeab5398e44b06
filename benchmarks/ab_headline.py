#!/usr/bin/env python3
"""A/B timing of builds of libomcmc_hip.so on the headline sweep: interleaved rounds on ONE box, one process
per (round, build); prints per-build min / median kernel time.

    python3 benchmarks/ab_headline.py name=path/to/lib.so [name=...] [--rounds 3] [--steps 200] [-- extra bench.py flags]

`name=-` means the in-tree build.  Kernel time = HIP events around the K launches of omc_gmrf_run (bench.py).
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    rounds, steps, builds = 3, 200, []
    it = iter(args)
    for a in it:
        if a == "--rounds":
            rounds = int(next(it))
        elif a == "--steps":
            steps = int(next(it))
        else:
            name, path = a.split("=", 1)
            builds.append((name, path))
    res = {n: [] for n, _ in builds}
    for r in range(rounds):
        for name, path in builds:
            env = dict(os.environ)
            if path != "-":
                env["OMC_HIP_LIB"] = os.path.join(ROOT, path)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "20", "--no-cpu"] + extra,
                                 env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(name, "FAILED", out.stderr[-400:], flush=True)
                continue
            j = json.loads(line[-1])
            res[name].append(1e3 * j["roofline"]["kernel_ms"])
            print(f"round {r} {name:12s} kernel {res[name][-1]:8.2f} us   step {1e3 * j['ms_per_step']:8.2f} us   lam {j['config']['check']['mean_lambda']:.4f}", flush=True)
    for name, v in res.items():
        if v:
            print(f"{name:12s} min {min(v):8.2f}  median {statistics.median(v):8.2f} us  ({len(v)} runs)")


if __name__ == "__main__":
    main()
