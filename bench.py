"""Headline benchmark: chain-updates/sec on the GMRF smoother (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W

For N > 1 the ranks are one process per GPU under torch.distributed.run (RCCL); started directly with
--gpus N > 1 (no WORLD_SIZE in the environment) the script launches that itself as a child process, before
anything has touched the GPU, and exits with the child's code.

Workload: 10 000 latent nodes, RW1 prior precision lambda*P, Gaussian
likelihood tau*I, samplers [NormalNormal(b), NormalGamma(lambda), NormalGamma(tau)], 1024 chains
per GPU.  One step = one full sweep of all samplers for every chain plus the per-iteration
bookkeeping of MCMC.run_mcmc (store of b/lambda/tau and log_post, mcmc.py:105-108); draws come
from the in-kernel Philox stream; all inputs are resident in HBM before the timed region.

Chains are independent, so ranks share nothing during sampling; the only collective is the gather of
the per-chain traces and stores at the end, outside the timed region.  --scaling strong (default, the
headline `value`; BASELINE.json's metric is "1024 chains ... at 1/2/4/8 GPUs", BASELINE.md section 3 "chains
sharded evenly"): 1024 chains in all.  --scaling weak: every GPU runs 1024 chains.  For N > 1 the default run
measures the other mode as well, after the headline, and reports it in the same JSON line (`other_scaling`).

Every timed run explains itself (`config.diagnostics`): the library's launch log (host clock around every kernel
launch of the run), the in-kernel sweep clock (entry and exit of every (sweep, chain) workgroup on the device's
constant-rate counter) and the deltas of the slow-path counters, reduced to: when the host issued, when the device
started, how long the sweeps took (median, maximum, which chain), whether slow sweeps were single chains or the whole
device at once -- so a slow run says where its time went.  After the headline and its repeats the other BASELINE
configs run for a bounded time each and are attached under `config.secondary` (never part of `value`).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NODES = 10000
CHAINS_PER_GPU = 1024
ALG_BYTES_PER_CHAIN_UPDATE = 40 * N_NODES  # d,t written+read (32n) + x written (8n)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
STORE_SLABS_MAX = 256  # stored iterations kept resident (ring); 82 MB each at 1024 x 10 000


def gmrf_problem(n):
    rng = np.random.default_rng(0)
    t = np.arange(n) * 60.0 / n
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
    d = np.full(n, 2.0)
    d[0] = d[-1] = 1.0
    d[0] += 1e-3
    return y, d, -np.ones(n - 1)


class GmrfSweep:
    """The cfg3 sweep driven through the C ABI (openmcmc_amd.engine)."""

    A_LAM, B_LAM, A_TAU, B_TAU = 10.0, 1.0, 1.0, 1.0

    def __init__(self, n, chains, seed, chain_offset, device, n_store, fused=True, seg=0):
        from openmcmc_amd.engine import Engine

        self.n, self.C = n, chains
        self.eng = eng = Engine(chains, seed=seed, device=device, chain_id_offset=chain_offset)
        y, d, off = gmrf_problem(n)
        self.d_y, self.d_d, self.d_off = eng.to_device(y), eng.to_device(d), eng.to_device(off)
        self.lam, self.tau = eng.full((chains,), 100.0), eng.full((chains,), 1.0)
        self.terms = eng.tridiag_terms(
            [{"diag": self.d_d, "off": self.d_off, "scale": self.lam},
             {"rhs": self.d_y, "center": self.d_y, "scale": self.tau}], n)
        self.logdetP = eng.tridiag_logdet(n, self.d_d, self.d_off)
        self.logdetI = eng.zeros(1)
        self.quad = eng.empty(2, chains)
        self.n_store = n_store
        self.store_b = eng.empty(n_store, chains, n)  # store["b"], iteration-major slabs
        self.store_lam = eng.empty(n_store, chains)
        self.store_tau = eng.empty(n_store, chains)
        self.store_lp = eng.empty(n_store, chains)
        self.it = 0
        self.scratch = eng.empty(chains, n)
        self.fused = fused
        if seg:
            eng.set_option("tridiag_seg", seg)
        # one prepared gamma-block array per store slot (the store pointers differ per slot)
        self.blocks = [eng.gamma_blocks(
            [{"a0": self.A_LAM, "b0": self.B_LAM, "n_pos": n, "store": self.store_lam[s], "logdet": self.logdetP},
             {"a0": self.A_TAU, "b0": self.B_TAU, "n_pos": n, "store": self.store_tau[s], "logdet": self.logdetI}], 2)
            for s in range(n_store)]

    def step(self, kernel_events=None):
        if self.fused:
            return self.step_fused(kernel_events)
        eng, n, it = self.eng, self.n, self.it
        slot = it % self.n_store
        x = self.store_b[slot]  # the draw is written straight into its store slab
        if kernel_events is not None:
            kernel_events[0].record()
        eng.tridiag_sample_canonical(n, self.terms, x, z=None, draw_index=3 * it, quad_out=self.quad)
        if kernel_events is not None:
            kernel_events[1].record()
        eng.normal_gamma_update(self.A_LAM, self.B_LAM, n, self.quad[0], self.lam, draw_index=3 * it + 1)
        eng.normal_gamma_update(self.A_TAU, self.B_TAU, n, self.quad[1], self.tau, draw_index=3 * it + 2)
        lp = self.store_lp[slot]
        eng.scaled_gauss_logpdf(n, self.tau, self.logdetI, self.quad[1], lp)
        eng.scaled_gauss_logpdf(n, self.lam, self.logdetP, self.quad[0], lp, accumulate=True)
        eng.gamma_logpdf(self.lam, self.A_LAM, self.B_LAM, lp, accumulate=True)
        eng.gamma_logpdf(self.tau, self.A_TAU, self.B_TAU, lp, accumulate=True)
        self.store_lam[slot].copy_(self.lam)
        self.store_tau[slot].copy_(self.tau)
        self.it += 1

    def run_fused(self, k):
        """k sweeps issued by ONE call into the library (omc_gmrf_run): no host work between launches."""
        if k <= 0:
            return
        eng, n = self.eng, self.n
        if getattr(self, "_run_args", None) is None:
            # the argument block of omc_gmrf_run, marshalled ONCE: a C caller passes these pointers as they are, and 35 us of
            # Python in front of every call (a tenth of a 20-sweep run at 128 chains per GPU) are not part of a sweep
            import ctypes as C

            from openmcmc_amd import _abi

            blocks = [{"a0": self.A_LAM, "b0": self.B_LAM, "n_pos": n, "store": self.store_lam, "logdet": self.logdetP, "draw_index": 1},
                      {"a0": self.A_TAU, "b0": self.B_TAU, "n_pos": n, "store": self.store_tau, "logdet": self.logdetI, "draw_index": 2}]
            B = eng._gamma_blocks_strided(blocks, self.terms.n_terms)
            self._run_args = (_abi.lib.omc_gmrf_run, _abi.check, eng._ctx, C.byref(self.terms), B, eng._p(self.store_b),
                              self.store_b.stride(1), self.store_b.stride(0), eng._p(self.store_lp), eng._p(self.scratch, self.C, n))
        fn, check, ctx, T, B, xs, ld_x, slot_stride, lp, scratch = self._run_args
        check(fn(ctx, n, T, B, 0, k, 1, 3 * self.it, 3, xs, ld_x, slot_stride, self.it % self.n_store, self.n_store, lp, scratch))
        self.it += k

    def step_fused(self, kernel_events=None):
        """The same sweep as ONE launch (omc_gmrf_sweep): draw, both Normal-Gamma updates, log_post,
        and the store writes (x, lambda, tau, log_post go straight into their store slots)."""
        eng, n, it = self.eng, self.n, self.it
        slot = it % self.n_store
        blocks = self.blocks[slot]
        if kernel_events is not None:
            kernel_events[0].record()
        eng.gmrf_sweep(n, self.terms, blocks, self.store_b[slot], z=None, draw_index=3 * it,
                       log_post_out=self.store_lp[slot], gamma_draw_base=3 * it + 1)
        if kernel_events is not None:
            kernel_events[1].record()
        self.it += 1


def cpu_baseline(n, seconds_budget=12.0):
    """Oracle (CPU restatement of the reference's sparse route: SuperLU factor + spsolve, same
    call pattern as gmrf.py) timed on this host, 1 chain, bounded sample."""
    from scipy import sparse

    from oracle import c_ref, sweep_ref

    y, d, off = gmrf_problem(n)
    P = sparse.diags((off, d, off), offsets=[-1, 0, 1], format="csc")
    rng = np.random.default_rng(1)
    k = 4
    t0 = time.perf_counter()
    sweep_ref.gmrf_smoother_chain(y, P, 0, k, rng.standard_normal((k, n)), 1 + rng.random((k, 2)))
    per = (time.perf_counter() - t0) / k
    k = int(max(8, min(2000, seconds_budget / per)))
    z, g = rng.standard_normal((k, n)), rng.standard_gamma(5000.0, size=(k, 2))
    t0 = time.perf_counter()
    sweep_ref.gmrf_smoother_chain(y, P, 0, k, z, g)
    dt = time.perf_counter() - t0
    # the plain-C Thomas restatement of the same sweep, for an un-flattering comparison
    mu = np.zeros(n)
    kc = 2000
    zc = rng.standard_normal((8, n))
    lam, tau = 100.0, 1.0
    t1 = time.perf_counter()
    for i in range(kc):
        _, q, _ = c_ref.tridiag_draw(d, off, lam, tau, y, mu, zc[i % 8])
    dtc = time.perf_counter() - t1
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    blas = None
    try:
        from threadpoolctl import threadpool_info

        blas = [{"api": i.get("internal_api"), "threads": i.get("num_threads")} for i in threadpool_info()]
    except Exception:
        pass
    ratio = None
    try:  # measured in the build container, reference and oracle on the same cores (tests/golden/make_golden.py)
        ratio = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_timing.json"))).get("oracle_over_reference_speed")
    except (OSError, ValueError):
        pass
    return {
        "value": k / dt, "unit": "chain-updates/s", "cores": 1, "kind": "port",
        "oracle_over_reference_speed": ratio,
        "reference_equivalent": (k / dt / ratio) if ratio else None,
        "reference_equivalent_note": "value / (oracle speed over the imported reference's on one host, "
                                     "tests/golden/reference_timing.json): what openMCMC itself would do on these cores",
        "cpu_model": cpu_model, "os_cpu_count": os.cpu_count(), "blas_threadpools": blas,
        "threads_note": "the sparse route (SuperLU factor + solves) and the C Thomas sweep are single-threaded; BLAS pools are idle here",
        "parallel_projection": k / dt * (os.cpu_count() or 1),
        "parallel_projection_note": "value x os.cpu_count(): every host core running its own chain (the reference "
                                    "itself has no multi-chain mode)",
        "sample": f"{k} sweeps of 1 chain (n={n}) of the oracle's sparse-route restatement "
                  f"(scipy SuperLU), draws pre-generated; host has {os.cpu_count()} cpus",
        "c_thomas_value": kc / dtc,
        "c_thomas_note": "oracle/c sequential O(n) restatement, draw only (no RNG, no log_post), 1 core",
    }


def secondary_lines(args):
    """cfg2 / cfg4 / cfg5 (benchmarks/secondary.py) for about --secondary-ms each, reduced to {ms_per_step, value, frac,
    check, cpu_baseline}; a config that fails reports its error and the headline line stays."""
    import types

    import torch

    from benchmarks import secondary

    out = {}
    # steps per config from the rates of profiles/ (cfg2 0.23 ms, cfg4 0.02 ms, cfg5 0.9 ms per step): about secondary_ms each
    plan = {"cfg2": (0.25, 5), "cfg4": (0.03, 20), "cfg5": (1.0, 3)}
    for name, (ms_guess, warm) in plan.items():
        steps = int(max(20, min(20000, args.secondary_ms / ms_guess)))
        a = types.SimpleNamespace(steps=steps, warmup=warm, chains=None, no_cpu=args.no_cpu, mala_products=False, mala_single=False, config=name,
                                  condition_ms=min(200.0, args.secondary_ms / 4))
        t0 = time.perf_counter()
        try:
            line = getattr(secondary, name)(a, torch)
            roof = line.get("roofline") or {}
            out[name] = {"metric": line["metric"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
                         "steps": steps, "warmup": warm, "workload": line["config"]["workload"], "check": line["config"].get("check"),
                         "roofline": {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "note", "factorisation_route",
                                                                "equivalent_tflops_on_reference_count")} if roof else None,
                         "cpu_baseline": line.get("cpu_baseline"), "route": line["config"].get("route"),
                         "seconds_spent": None}
        except Exception as exc:  # noqa: BLE001 -- the headline must survive
            out[name] = {"error": repr(exc)}
        out[name]["seconds_spent"] = time.perf_counter() - t0
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=["cfg3", "cfg2", "cfg4", "cfg5"], default="cfg3",
                    help="cfg3 (default): the headline GMRF smoother; the others: benchmarks/secondary.py, one GPU")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (weak) / in all (strong); default: the config's")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default; BASELINE's metric): --chains in all, sharded evenly over the GPUs; weak: --chains per GPU")
    ap.add_argument("--one-mode", action="store_true", help="N > 1: do not measure the other scaling mode as well")
    ap.add_argument("--condition-sweeps", type=int, default=4096,
                    help="untimed device conditioning before the W warm-up steps: this many sweeps of the same kernel on the same "
                         "chains (a FIXED count: state, draw indices and check values are the same on every box), so that a short "
                         "run does not time the clock ramp out of the idle power state (first ~10 ms of load); ends with a dress "
                         "rehearsal of the timed run (event pair + K sweeps + read-back of the diagnostics)")
    ap.add_argument("--secondary-ms", type=float, default=1000.0,
                    help="N = 1: after the headline, run cfg2 / cfg4 / cfg5 for about this long each and attach them under "
                         "config.secondary (0 = skip)")
    ap.add_argument("--repeat-ms", type=float, default=250.0,
                    help="after the headline, repeat the K-step run until this much time has been measured (spread report)")
    ap.add_argument("--collective-timeout", type=float, default=240.0,
                    help="N > 1: seconds the gathers and the second scaling mode may take after the headline before rank 0 prints "
                         "the line without them and all ranks leave (0 = wait for ever)")
    ap.add_argument("--no-library-gather", action="store_true",
                    help="N > 1 on GPUs: do not time the store gather through the library's own RCCL collective as well")
    ap.add_argument("--nodes", type=int, default=N_NODES)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-sweep-clock", action="store_true", help="diagnostic: without the in-kernel sweep clock (one 16-byte store per workgroup and sweep)")
    ap.add_argument("--python-loop", action="store_true", help="issue every sweep from Python instead of omc_gmrf_run")
    ap.add_argument("--unfused", action="store_true", help="one launch per sampler instead of the fused sweep kernel")
    ap.add_argument("--zero-z", action="store_true", help="diagnostic what-if: no draw generation (results are not samples)")
    ap.add_argument("--generic", action="store_true", help="diagnostic: never use the structure-specialised kernel instantiation")
    ap.add_argument("--stamps", action="store_true", help="diagnostic: print in-kernel phase stamps (not a timing run)")
    ap.add_argument("--seg", type=int, default=0, help="nodes per lane of the segmented kernel (0 = auto)")
    ap.add_argument("--mala-products", action="store_true",
                    help="cfg4: the step as dense products (omc_mala_step) instead of the whitened step")
    ap.add_argument("--mala-single", action="store_true", help="cfg4: one launch pair per step (omc_mala_step_white) instead of blocks of steps")
    ap.add_argument("--reenter", type=int, default=None, choices=[0, 1, 2],
                    help="omc_gmrf_run: 1 = workgroups restart themselves for the next sweep of a launch (default: the library's)")
    ap.add_argument("--sweeps-per-launch", type=int, default=0, help="omc_gmrf_run: sweeps per launch (0 = library default, 32)")
    ap.add_argument("--block-sweeps", type=int, default=None,
                    help="omc_gmrf_run: sweeps a self-restarting workgroup walks before a fresh one takes its chain over (0 = the whole launch)")
    args = ap.parse_args()

    if args.config != "cfg3":
        from benchmarks import secondary

        if args.steps == 200 and args.config == "cfg5":
            args.steps, args.warmup = 20, 3
        return secondary.main(args)
    if args.chains is None:
        args.chains = CHAINS_PER_GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started directly: become the launcher.  Nothing has touched the GPU yet (torch is not even imported), and
        # the ranks are CHILD processes -- never an exec from a process that holds the GPU.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("OMC_BENCH_BACKEND", "nccl") != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    torch.cuda.set_stream(torch.cuda.Stream())  # a real stream: the legacy default stream serialises with everything
    dist = None
    if world > 1:
        import torch.distributed as dist

        # RCCL over xGMI in production; OMC_BENCH_BACKEND=gloo rehearses the same launch contract where
        # ranks have to share one GPU (LOCAL_RANK is then folded onto the visible devices)
        backend = os.environ.get("OMC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    n = args.nodes
    on_gpu = dist is None or dist.get_backend() == "nccl"

    def barrier():
        if dist is not None:
            dist.barrier()

    def shard(mode):
        """(chains of this rank, global id of its first chain) -- contiguous blocks; strong: as even as it goes"""
        if mode == "weak":
            return args.chains, rank * args.chains
        base, extra = divmod(args.chains, world)
        return base + (1 if rank < extra else 0), rank * base + min(rank, extra)

    def diagnose(sweep, snap, t_host0, ev_ms, counters0, counters1, launches):
        """Where did the time of one timed K-sweep run go?  From the library's launch log (host clock), the sweep clock
        (device counter at entry / exit of every (sweep, chain) workgroup; `snap` = the run's records, copied on the device
        right after the run) and the slow-path counters.  Called after ALL timed runs: a read-back and a few milliseconds
        of numpy between two runs leave the device idle long enough for its clocks to drop (measured: the next run 13 %
        slow, the ones after it recovering over ~10 runs)."""
        eng = sweep.eng
        d = {}
        d["counters"] = {nm: counters1[nm] - counters0[nm] for nm in counters0}
        n_launch, recs = launches
        d["launches"] = n_launch
        d["host_issue_ms"] = [[round(1e3 * (r["t_begin"] - t_host0), 4), round(1e3 * (r["t_end"] - t_host0), 4)] for r in recs[:8]]
        d["launch_form"] = sorted({r["form"] for r in recs})
        if snap is None:
            return d
        tk = snap.cpu().numpy()  # (k, C, 2) ticks
        k = tk.shape[0]
        khz = float(eng.counter("wall_clock_khz")) or 1e5
        start, end = tk[:, :, 0].astype(np.float64), tk[:, :, 1].astype(np.float64)
        dur = (end - start) / khz  # ms per (sweep, chain)
        t_first = start.min()
        d["device_span_ms"] = float((end.max() - t_first) / khz)
        if ev_ms is not None:  # the part of the event interval in which no workgroup of the run was on the device
            d["outside_kernels_ms"] = float(ev_ms - d["device_span_ms"])
        med = float(np.median(dur))
        i_max = np.unravel_index(int(np.argmax(dur)), dur.shape)
        d["sweep_ms"] = {"median": med, "p99": float(np.percentile(dur, 99)), "max": float(dur.max()),
                         "max_at": {"sweep": int(i_max[0]), "chain": int(i_max[1])}}
        slow = dur > 3.0 * med
        n_slow = int(slow.sum())
        d["slow_sweeps"] = {"count": n_slow, "of": int(dur.size), "chains": int(slow.any(axis=0).sum())}
        # gaps between a chain's consecutive sweeps (self-restarting workgroups: none) and between launches
        if k > 1:
            gap = (start[1:] - end[:-1]) / khz
            d["gap_between_sweeps_ms"] = {"median": float(np.median(gap)), "max": float(gap.max())}
        if n_slow:
            # do the slow sweeps share an instant (a device-wide stall: every resident workgroup at once) or are they
            # single chains (a slow path of the kernel)?
            s_lo, s_hi = start[slow].max(), end[slow].min()
            d["slow_sweeps"]["share_an_instant"] = bool(s_lo < s_hi)
            d["slow_sweeps"]["window_ms"] = [float((start[slow].min() - t_first) / khz), float((end[slow].max() - t_first) / khz)]
            ch = np.nonzero(slow.any(axis=0))[0]
            d["slow_sweeps"]["first_chains"] = [int(c) for c in ch[:8]]
        return d

    def explain(d, ms_total, ms_typical):
        """One sentence for a run that took much longer than the typical one."""
        if ms_typical is None or ms_total < 1.5 * ms_typical:
            return None
        if d.get("host_issue_ms") and d["host_issue_ms"][0][0] > 0.5 * (ms_total - ms_typical):
            return "the host issued the first launch late (host_issue_ms): time lost on the CPU side before the kernel was queued"
        if "outside_kernels_ms" in d and d["outside_kernels_ms"] > 0.5 * (ms_total - ms_typical):
            return ("the device ran the sweeps at normal speed but started late or sat idle between the events (outside_kernels_ms): "
                    "queue or host, not the kernel")
        sl = d.get("slow_sweeps", {})
        if sl.get("count"):
            if sl.get("share_an_instant") and sl["chains"] >= min(64, sl["of"]):
                return ("every resident workgroup stalled over the same interval (slow_sweeps.window_ms): a device-wide stall from "
                        "outside the kernel (preemption, clock or power event), not a slow path of a chain")
            fb = d.get("counters", {}).get("tridiag_join_fallbacks", 0)
            return (f"{sl['chains']} chain(s) ran long sweeps" + (f"; {fb} pivot-join fallbacks in the run" if fb else "")
                    + ": a slow path of the kernel on those chains (slow_sweeps.first_chains)")
        return "no single cause in the record: sweeps uniformly slower (clock) -- compare sweep_ms.median with the repeats"

    def measure(mode, diagnostics=False):
        """One measurement in the driver's contract: W untimed warm-up steps, then exactly K steps between
        barrier + synchronize on both sides, MAX over ranks; then the same K-step run repeated (untimed in the
        headline) for the spread.  Every timed run leaves its diagnostics."""
        C, offset = shard(mode)
        n_store = max(1, min(args.steps, STORE_SLABS_MAX))
        sweep = GmrfSweep(n, C, seed=2025, chain_offset=offset, device=local, n_store=n_store,
                          fused=not args.unfused, seg=args.seg)
        if args.zero_z:
            sweep.eng.set_option("debug_zero_z", 1)
        if args.generic:
            sweep.eng.set_option("tridiag_generic", 1)
        if args.sweeps_per_launch:
            sweep.eng.set_option("run_sweeps_per_launch", args.sweeps_per_launch)
        if args.reenter is not None:
            sweep.eng.set_option("run_reenter", args.reenter)
        if args.block_sweeps is not None:
            sweep.eng.set_option("run_block_sweeps", args.block_sweeps)
        stamps = None
        if diagnostics and args.stamps:
            stamps = torch.zeros(C * 16 * 16, dtype=torch.int64, device="cuda")
            sweep.eng.set_option("stamps_ptr", stamps.data_ptr())
        c_loop = not (args.python_loop or args.unfused)
        ring = None
        if c_loop and not args.no_sweep_clock:
            ring = sweep.eng.sweep_clock(max(64, 32 * ((args.steps + 31) // 32) + 32))
        counter_names = ["tridiag_join_fallbacks", "run_handoff_timeouts"]

        def run_k(events):
            if c_loop:  # one library call issues all K launches; the events bracket them on the launch stream -- recorded by
                #         the library itself around its launches (options run_event_begin / run_event_end, set below): two
                #         interpreter-level event calls cost more than the launch and are no part of a step
                sweep.run_fused(args.steps)
            else:
                for i in range(args.steps):
                    sweep.step(events[i] if events else None)

        use_ev = not args.no_kernel_events
        n_ev = 1 if c_loop else args.steps
        # the timing events exist and have been recorded once before any timed window opens
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)] if use_ev else None
        if events:
            for a, b in events:
                a.record()
                b.record()
            torch.cuda.synchronize()
            if c_loop:  # the raw handles exist now; omc_gmrf_run records them on its stream from here on
                sweep.eng.set_option("run_event_begin", events[0][0].cuda_event)
                sweep.eng.set_option("run_event_end", events[0][1].cuda_event)

        ring_idx = None
        stream_now = torch.cuda.current_stream()  # (the launch stream: fixed for the process, looked up once)

        def timed():
            """-> (seconds, MAX over ranks; kernel ms per sweep from the events; raw material of the run's diagnostics)"""
            nonlocal ring_idx
            counters0 = {nm: sweep.eng.counter(nm) for nm in counter_names} if c_loop else {}
            pos0 = sweep.eng.counter("sweep_times_pos") if ring is not None else 0
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_k(events)
            # (a blocking synchronize sleeps and is woken some 20 us after the last kernel has retired -- 1.3 % of a 20-sweep
            # run; the stream is polled first, the synchronize of the contract then returns at once)
            while not stream_now.query():
                pass
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0  # this rank's K steps; the MAX over ranks below is the job's time
            barrier()                      # (the closing barrier of the bracket: an RCCL barrier is a collective launch of
                                           #  tens of microseconds -- a tenth of a 20-step run at 128 chains -- and is not a step)
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            if dist is not None:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            ev_total = float(np.sum([a.elapsed_time(b) for a, b in events])) if use_ev else None
            kern = ev_total / args.steps if use_ev else None
            raw = {"t0": t0, "ev_ms": ev_total, "wall_ms": 1e3 * dt, "counters0": counters0}
            if c_loop:  # nothing here waits for the device or keeps it idle for long: the analysis comes after all runs
                raw["launches"] = sweep.eng.launch_log()
                if ring is not None:
                    if ring_idx is None or ring_idx[0] != pos0:  # (index vector, ring position of its first entry)
                        ring_idx = [pos0, (torch.arange(args.steps, device="cuda") + pos0) % ring.shape[0]]
                    raw["snap"] = ring[ring_idx[1]]  # device-side copy of the run's records
                    ring_idx = [(pos0 + args.steps) % ring.shape[0], (ring_idx[1] + args.steps) % ring.shape[0]]
            return tmax.item(), kern, raw

        def finish(raws):
            """the diagnostics of the runs, in order (a run's counter deltas end where the next run's start)"""
            out = []
            last = {nm: sweep.eng.counter(nm) for nm in counter_names} if c_loop else {}
            for i, r in enumerate(raws):
                c1 = raws[i + 1]["counters0"] if i + 1 < len(raws) else last
                g = diagnose(sweep, r.get("snap"), r["t0"], r["ev_ms"], r["counters0"], c1, r["launches"]) if c_loop else {}
                g["wall_ms"], g["events_ms"] = r["wall_ms"], r["ev_ms"]
                out.append(g)
            return out

        # Device conditioning (untimed, before the contract's W warm-up steps): an idle MI355X needs ~10 ms of load to
        # leave its low-power clocks (measured: the first 100 sweeps after setup run at 91.9 us, every later 100 at
        # 85.4 us, profiles/README.md r02a); a --steps 20 run would time nothing but that ramp.  The chains simply run
        # longer burn-in: same kernel, same buffers (the store ring is touched once through, like the reference's
        # NaN-filled store arrays are before its loop, mcmc.py:88-95).  A FIXED number of sweeps, so that the chains'
        # state, every draw index and the check values are the same on every box and in every run; the last K of them
        # are a dress rehearsal of the timed run (same call, same events, same read-back).
        rehearsal = None
        if c_loop and args.condition_sweeps > 0 and not (diagnostics and args.stamps):
            # First everything the timed run's bookkeeping touches for the first time in this process (the events' first
            # timed interval, the counters' copies, torch's index kernel for the sweep-clock snapshot: lazily loaded code
            # objects cost milliseconds of host time with the device idle, and an idle device drops its clocks within
            # milliseconds -- measured: a timed run that follows such a gap starts 12 % slow and recovers over ~20 ms).
            # Then the conditioning proper, with no idle gap between it and the timed runs.
            timed()
            left = max(0, args.condition_sweeps - 2 * args.steps)
            while left > 0:
                k = min(left, 256)
                sweep.run_fused(k)
                left -= k
                torch.cuda.synchronize()
            _, rk, rd = timed()
            rehearsal = {"kernel_ms": rk, "wall_ms": rd["wall_ms"]}
            del rd
        if c_loop:
            sweep.run_fused(args.warmup)
        else:
            for _ in range(args.warmup):
                sweep.step()
        sweep.eng.check_status()

        dt, kern_ms, raw0 = timed()
        sweep.eng.check_status()
        # spread: the same K-step run again (every rank the same count: the loop bound comes from rank-agreed dt)
        reps = int(min(20, max(0, np.ceil(args.repeat_ms * 1e-3 / max(dt, 1e-6)) - 1))) if not (diagnostics and args.stamps) else 0
        rep_ms, rep_kern, raws = [1e3 * dt / args.steps], [kern_ms], [raw0]
        for _ in range(reps):
            d, k, rw = timed()
            rep_ms.append(1e3 * d / args.steps)
            rep_kern.append(k)
            raws.append(rw)
        sweep.eng.check_status()
        rep_diag = finish(raws)
        diag = rep_diag[0]
        del raws, raw0
        spread = {"runs": len(rep_ms), "ms_per_step_min": min(rep_ms), "ms_per_step_median": float(np.median(rep_ms)),
                  "ms_per_step_max": max(rep_ms)}
        if use_ev:
            spread.update(kernel_ms_min=min(rep_kern), kernel_ms_median=float(np.median(rep_kern)), kernel_ms_max=max(rep_kern))
        # every run against the typical one; a run that stands out carries its explanation
        typical = float(np.median([g["wall_ms"] for g in rep_diag])) if len(rep_diag) > 1 else (rehearsal["wall_ms"] if rehearsal else None)
        for g in rep_diag:
            why = explain(g, g["wall_ms"], typical)
            if why:
                g["verdict"] = why
        outliers = [dict(run=i, **g) for i, g in enumerate(rep_diag) if i > 0 and "verdict" in g]
        report = {"headline_run": diag, "rehearsal": rehearsal, "typical_wall_ms": typical,
                  "repeats_with_a_verdict": outliers[:4],
                  "sweep_ms_median_per_run": [g.get("sweep_ms", {}).get("median") for g in rep_diag],
                  "reenter_abi_ok": sweep.eng.counter("reenter_abi_ok") if c_loop else None}
        return {"mode": mode, "C": C, "dt": dt, "kern_ms": kern_ms, "sweep": sweep, "stamps": stamps, "spread": spread,
                "n_store": n_store, "diagnostics": report}

    def make_line(lam_mean, trace_error, gather_info, other):
        """the bench line of this run (rank 0), with whatever of the collective part is known"""
        kern_ms = m["kern_ms"]
        total_chains = args.chains * world if args.scaling == "weak" else args.chains
        value = total_chains * args.steps / dt
        out = {
            "metric": "chain-updates/sec (1024 chains, 10k-node GMRF) at 1/2/4/8 GPUs vs CPU ref",
            "value": value, "unit": "chain-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"GMRF smoother (examples/4): {n} nodes, RW1 precision, "
                                   + (f"{args.chains} chains per GPU, " if args.scaling == "weak" else f"{args.chains} chains in all, sharded evenly, ")
                                   + "NormalNormal + 2x NormalGamma + store + log_post per step"
                                   + (" (one fused launch)" if not args.unfused else " (one launch per sampler)"),
                       "chains_total": total_chains, "chains_rank0": C, "nodes": n, "parallelism": f"chains sharded x{world}",
                       "check": {"mean_lambda": lam_mean, **({"trace_gather_error": trace_error} if trace_error else {})},
                       "store_gather": gather_info,
                       "repeats": m["spread"], "other_scaling": other,
                       "diagnostics": m["diagnostics"]},
        }
        if kern_ms is not None:
            # One launch of omc_gmrf_run carries up to 32 sweeps (blocks = sweeps x chains): per launch the kernel
            # processes spl x C chain-updates.  `kernel_ms` stays the time per SWEEP (events around all launches of the
            # timed region / K); the launch figures are those of a full launch.
            spl = args.sweeps_per_launch or 32
            spl = 1 if (args.python_loop or args.unfused) else min(spl, args.steps)
            alg_sweep = ALG_BYTES_PER_CHAIN_UPDATE * (n / N_NODES) * C
            achieved = alg_sweep / (kern_ms * 1e-3) / 1e9
            traffic, rec = None, {}
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                rec = json.load(open(tpath))
                if rec.get("nodes") == n and rec.get("chains") == C:
                    traffic = rec.get("hbm_bytes_per_sweep") * spl
                else:
                    rec = {}
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": "k_tridiag_seg", "kernel_ms": kern_ms,
                               "sweeps_per_launch": spl, "launch_ms": kern_ms * spl,
                               "alg_bytes_per_launch": alg_sweep * spl,
                               "note": "achieved = alg_bytes_per_launch / launch_ms (SURVEY 8d's 40 n B per chain-update, the contract's "
                                       "figure); traffic = PMC HBM bytes per launch (profiles/traffic.json, per sweep x "
                                       "sweeps_per_launch).  The chain stays on chip: the bytes that really cross the HBM pins are in "
                                       "hbm_measured, and what limits the kernel is in roofline_valu_issue"}
            if traffic:
                # the rate at the HBM pins: counter bytes / this run's launch time (north_star's 'rocprof HBM GB/s')
                gbs = traffic / (kern_ms * spl * 1e-3) / 1e9
                out["roofline"]["hbm_measured"] = {"GBps": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS,
                                                   "bytes_per_launch": traffic, "source": rec.get("source")}
            if rec.get("valu_instructions_per_wave_and_sweep"):
                # The bound the counters point at: vector-ALU issue.  One workgroup (16 waves, 4 per SIMD) owns a CU for a
                # chain's sweep; the C chains take ceil(C / CUs) rounds.  issue cycles of a workgroup-sweep on one SIMD =
                # instructions per wave x waves per SIMD x measured cycles per instruction (benchmarks/micro/valu_rates.hip);
                # available = the sweep's share of the launch in core cycles.
                cus, ghz = 256, float(rec.get("core_clock_ghz", 2.4))
                rounds = -(-C // cus)
                cyc_avail = kern_ms * 1e-3 * ghz * 1e9 / rounds
                ipw, cpi = float(rec["valu_instructions_per_wave_and_sweep"]), float(rec.get("cycles_per_valu_instruction", 5.1))
                issue = ipw * 4 * cpi
                out["roofline_valu_issue"] = {
                    "bound": "valu_issue", "achieved": issue, "peak": cyc_avail, "unit": "SIMD cycles per workgroup-sweep",
                    "frac": issue / cyc_avail, "valu_instructions_per_wave_and_sweep": ipw, "waves_per_simd": 4,
                    "cycles_per_instruction": cpi, "core_clock_ghz": ghz, "rounds_of_workgroups": rounds,
                    "source": rec.get("source"),
                    "note": "share of a workgroup-sweep's time its SIMDs spend issuing vector instructions (counter instruction "
                            "count x cycles per fp64/int instruction measured at 4 waves per SIMD); the rest is latency at the "
                            "phase barriers of a one-workgroup-per-CU design"}
        return out

    m = measure(args.scaling, diagnostics=True)
    sweep, C, dt, n_store, stamps = m["sweep"], m["C"], m["dt"], m["n_store"], m["stamps"]

    # The headline is measured.  What follows for N > 1 -- the library's own RCCL communicator, the gathers, the other scaling
    # mode -- has never met real multi-GPU hardware in this repository (one-GPU boxes; gloo rehearsals only), and a collective
    # that hangs would take the measured line with it.  A watchdog therefore holds the line as it stands now: if the rest
    # is not through after --collective-timeout seconds, rank 0 prints it (saying so in config.store_gather) and every rank
    # leaves; a rank stuck inside a native call cannot be interrupted any other way.
    watchdog = None
    fallback_line = {}
    if dist is not None and args.collective_timeout > 0:
        import threading

        def give_up():
            if rank == 0 and fallback_line:
                fallback_line["config"]["store_gather"] = {"error": f"not finished after {args.collective_timeout:.0f} s: gave up (watchdog)"}
                print(json.dumps(fallback_line), flush=True)
            # a collective that hangs is a defect: the measured line is out, but the job must not read as a success
            print(f"rank {rank}: collective part not finished after {args.collective_timeout:.0f} s (watchdog), exiting 3",
                  file=sys.stderr, flush=True)
            os._exit(3)

        watchdog = threading.Timer(args.collective_timeout, give_up)
        watchdog.daemon = True

    # the one collective of the path: gather of the per-chain traces (outside the timed region)
    trace = torch.stack([sweep.store_lam[: min(args.steps, n_store)], sweep.store_tau[: min(args.steps, n_store)]])
    if not on_gpu:
        trace = trace.cpu()
    trace_error = None
    comm, collective = None, None
    if watchdog is not None:
        if rank == 0:
            fallback_line.update(make_line(trace[0].mean().item(), "traces of rank 0's chains only (watchdog line)", None, None))
        watchdog.start()
    if dist is not None:
        # torch.distributed's gather on the group's backend (RCCL under the driver, gloo in rehearsals): the collective every
        # check value and the reported store gather rest on.  The library's own RCCL collective (omc_gather_samples: every peer
        # sends point to point into the root) is timed on the same block AFTER these, so that a problem in it -- it has met
        # several GPUs in no box of this pool yet -- cannot take them with it (config.store_gather.library).
        collective = "torch.distributed.gather (" + dist.get_backend() + ")"
        try:
            from openmcmc_amd.parallel import gather_chains

            trace = gather_chains(trace, chain_dim=2, dst=0, comm=None)
        except Exception as exc:  # the check value then covers rank 0's chains only; the headline stays
            trace_error = repr(exc)
    lam_mean = trace[0].mean().item() if rank == 0 else None

    # ... and of the sample store itself: timed on a bounded part (the last <= 8 stored iterations of b,
    # 82 MB each per rank at 1024 chains) so the run stays short; reported next to the headline, never inside it
    gather_info = None
    if dist is not None:
        try:
            from openmcmc_amd.parallel import gather_chains

            k_it = min(8, n_store)
            part = sweep.store_b[:k_it].contiguous()
            if not on_gpu:
                part = part.cpu()
            barrier()
            torch.cuda.synchronize()
            tg = time.perf_counter()
            full = gather_chains(part, chain_dim=1, dst=0, comm=None)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            nbytes = (full.numel() - part.numel()) * 8 if rank == 0 else 0
            gather_info = {"collective": collective, "iterations": k_it, "bytes_into_root": nbytes, "ms": 1e3 * tg,
                           "GBps_into_root": nbytes / tg / 1e9}
            ref_sum = full.sum().item() if rank == 0 else None
            del full
            if rank == 0 and fallback_line:  # what is known so far goes into the line the watchdog would print
                fallback_line["config"]["check"]["mean_lambda"] = lam_mean
                fallback_line["config"]["store_gather_before_watchdog"] = dict(gather_info)
            if on_gpu and not args.no_library_gather:
                # the same block once more through the library's own collective
                try:
                    from openmcmc_amd.parallel import make_communicator

                    comm = make_communicator(sweep.eng)
                    barrier()
                    torch.cuda.synchronize()
                    tg = time.perf_counter()
                    full = gather_chains(part, chain_dim=1, dst=0, comm=comm)
                    torch.cuda.synchronize()
                    tg = time.perf_counter() - tg
                    gather_info["library"] = {"collective": "omc_gather_samples (RCCL send/recv into the root)", "ms": 1e3 * tg,
                                              "GBps_into_root": nbytes / tg / 1e9,
                                              "same_result": bool(full.sum().item() == ref_sum) if rank == 0 else None}
                    del full
                except Exception as exc:
                    gather_info["library"] = {"error": repr(exc)}
        except Exception as exc:  # the headline must survive a collective problem
            gather_info = {"error": repr(exc)}

    if stamps is not None and rank == 0:
        st = stamps.cpu().numpy().reshape(C, 16, 16).astype(np.float64)
        if os.environ.get("OMC_STAMPS_DUMP"):  # raw [chain][wave][stamp] ticks for a timeline (benchmarks/stamp_timeline.py)
            np.save(os.environ["OMC_STAMPS_DUMP"], st)
        nwv = int((st[0, :, 0] > 0).sum())
        d = np.diff(st[:, :nwv, :], axis=2)
        names = ["gamma draws", "fill b", "fill a", "moebius local", "moebius scan", "newton", "l", "fill rhs",
                 "fwd map+scan", "fwd pass+rng", "bwd map+scan", "bwd pass", "tile+quad", "quad sum", "epilogue+store"]
        tot = (st[:, :nwv, 15] - st[:, :nwv, 0]).mean()
        print("phase stamps, %% of a wave's lifetime, mean over chains and waves; wave 0 and wave %d shown" % (nwv - 1), file=sys.stderr)
        for i, nm in enumerate(names):
            print(f"  {nm:16s} mean {100 * d[:, :, i].mean() / tot:6.2f}   wave0 {100 * d[:, 0, i].mean() / tot:6.2f}   last {100 * d[:, -1, i].mean() / tot:6.2f}", file=sys.stderr)
        print(f"  wave lifetime {tot:.0f} ticks; block span {(st[:, :nwv, 15].max(1) - st[:, :nwv, 0].min(1)).mean():.0f} ticks", file=sys.stderr)

    # N > 1: the other scaling mode as well (a second, separate measurement after the headline)
    other = None
    if comm is not None:
        comm.close()
    if world > 1 and not args.one_mode and not args.stamps:
        del sweep
        m["sweep"] = None
        torch.cuda.empty_cache()
        o = measure("strong" if args.scaling == "weak" else "weak")
        tot_o = args.chains * world if o["mode"] == "weak" else args.chains
        other = {"scaling": o["mode"], "chains_total": tot_o, "chains_this_rank": o["C"],
                 "value": tot_o * args.steps / o["dt"], "unit": "chain-updates/s", "ms_per_step": 1e3 * o["dt"] / args.steps,
                 "kernel_ms": o["kern_ms"], "repeats": o["spread"]}
        o["sweep"] = None

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out = make_line(lam_mean, trace_error, gather_info, other)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(n)
        if world == 1 and args.secondary_ms > 0 and not args.stamps:
            # the other BASELINE configs, bounded, after everything the headline needs has been measured
            m["sweep"] = None
            del sweep
            torch.cuda.empty_cache()
            out["config"]["secondary"] = secondary_lines(args)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
