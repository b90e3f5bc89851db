#!/usr/bin/env python3
"""bench.py -- prox! throughput of the MI355X-native shifted proximal operators.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
    ShiftedNormL1Box prox!, n = 10^8 fp64, Delta = 1.0 (scalar bounds l = -1, u = 1, every index selected),
    twice shifted (xk ~ N(0,1), sj ~ U(-1/2, 1/2)), q ~ N(0,1), lambda = sigma = 1.  Synthetic data.
A "step" is one prox! call over the whole n-vector.  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--elements N] [--no-cpu] [--no-extra]

Multi-GPU (launched by torch.distributed.run, one rank per GPU): the operators are separable, so every rank
owns its own contiguous n-element shard and there is NO data-path collective ("replicas / weak scaling",
SURVEY.md 8e); the only collectives are the barrier and a MAX over ranks of the elapsed time.

One JSON line on rank 0: metric/value/... plus
  roofline      -- dominant kernel (k_sep_lds<OpL1Box>): algorithmic bytes (32 B/element) / average launch
                   duration measured with HIP events on the launching stream over the timed region
  cpu_baseline  -- the CPU oracle (literal single-thread C restatement of the reference's Julia loop; the
                   reference itself cannot run here: no Julia) timed on rank 0's host core
  other_operators         -- the other BASELINE configs and the adjacent calls (iprox!, psi(y), prox! + psi in one pass,
                             ShiftedNormL1B2, vector bounds, the PCIe-inclusive host-pointer form) at full size
  cpu_port_other_configs  -- the single-thread CPU port on bounded samples of the other configs
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
BYTES_PER_ELEM = 32    # read q, xk, sj + write y (scalar bounds, no mask)
SYNTH_SEED = 20250613  # spx_synth_fill seed (+ rank); streams: 0 = xk, 1 = sj, 2 = q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--elements", dest="n", type=int, default=100_000_000, help="elements per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary operators")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one per GPU, RCCL) as a child
        # process -- before anything here touches the GPU -- and leave with its exit code
        import subprocess
        port = os.environ.get("MASTER_PORT", "29517")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; libspx has no CPU path")
    # SPX_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices,
    # barrier / MAX go over gloo); the driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("SPX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # SPX_BENCH_FORCE_DIST=1: initialise the process group, barrier and MAX-reduce also with ONE rank (under torch.distributed.run
    # --nproc-per-node 1) -- the RCCL branch of this file executed on the one-GPU box (profiles/r04_bench_nccl_1rank.txt)
    dist_on = world > 1 or (os.environ.get("SPX_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    s = ge.build()
    L = s._lib.load()

    n = args.n
    # SURVEY 8d's shared generator: spx_synth_fill(seed, stream, kind) on the device; the cpu_baseline leg regenerates the
    # same bits on the host (oracle.synth_fill) instead of copying them back
    ctx = s.context(dev)
    seed = SYNTH_SEED + rank
    xk, sj, q = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    for t, stream, kind in ((xk, 0, 1), (sj, 1, 0), (q, 2, 1)):
        s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), n, seed, stream, kind, 1.0))
    y = torch.empty_like(q)
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormLinf(1.0)), sj)

    def barrier():
        if dist_on:
            dist.barrier()

    # clock ramp: ~0.1 s of the same call so that the W warm-up steps and the K timed steps see steady-state
    # clocks (the GPU idles while the inputs are being generated); not counted anywhere
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.1:
        for _ in range(20):
            s.prox_bang(y, psi, q, 1.0)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        s.prox_bang(y, psi, q, 1.0)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s._lib.check(L.spx_timer_start(ctx))
    for _ in range(args.steps):
        s.prox_bang(y, psi, q, 1.0)
    ms = ctypes.c_float()
    s._lib.check(L.spx_timer_stop(ctx, ctypes.byref(ms)))  # HIP events on the launching stream
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([wall, ms.value], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(t[0]), float(t[1])
    else:
        ev_ms = float(ms.value)

    total_elems = float(n) * world * args.steps
    value = total_elems / wall / 1e9
    launch_ms = ev_ms / args.steps
    achieved = BYTES_PER_ELEM * n / (launch_ms * 1e-3) / 1e9

    out = {
        "metric": "prox! throughput (ShiftedNormL1Box, n=%.0e fp64 per GPU)" % n,
        "value": round(value, 3),
        "unit": "G-elements/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (spx_synth_fill: counter-based splitmix64 generator, seed %d + rank, streams 0/1/2: xk ~ N(0,1) "
                "[Irwin-Hall of 12], sj ~ U(-1/2,1/2), q ~ N(0,1); the CPU leg regenerates the same bits on the host and checks "
                "them against the device arrays by checksum; other_operators use the same generator, seed + 1000, streams 0..9)" % SYNTH_SEED,
        "config": {"workload": "ShiftedNormL1Box prox!, n=%d fp64 per GPU, Delta=1.0 scalar bounds, all selected, "
                               "twice shifted, lambda=sigma=1 (BASELINE configs[1])" % n,
                   "elements_per_gpu": n, "parallelism": "replicas (independent shards, no collective)"},
        "roofline": {"bound": "hbm", "kernel": "k_sep_lds<OpL1Box, 6, false, false>",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "avg_launch_ms": round(launch_ms, 5), "algorithmic_bytes_per_launch": BYTES_PER_ELEM * n,
                     **_pmc_traffic(n)},
    }

    if rank == 0 and not args.no_extra:
        out["other_operators"] = _extra(s, L, ctx, dev, n, torch)
        try:
            out["solver_iteration_in_a_graph"] = _graph_iteration(s, dev, torch)
        except Exception as e:  # reported, never fatal for the headline line
            out["solver_iteration_in_a_graph"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and not args.no_cpu:
        out["cpu_baseline"] = _cpu_baseline(s, psi, q, xk, sj, y, n, torch)
        if not args.no_extra:
            out["cpu_port_other_configs"] = _cpu_other_configs()
    barrier()
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


def _pmc_traffic(n):
    """HBM bytes per launch of the headline kernel.  NOT measured by this run: replayed from the committed rocprofv3 --pmc
    passes of this very command at the default size (tools/profile.sh -> profiles/traffic_l1box.json; FETCH_SIZE and
    WRITE_SIZE in separate passes, corrections as MI355X_MICROARCH.md prescribes); null for any other size."""
    p = os.path.join(ROOT, "profiles", "traffic_l1box.json")
    out = {"traffic": None, "traffic_source": None}
    if os.path.exists(p):
        try:
            d = json.load(open(p))
            if d.get("n") == n:
                out["traffic"] = d.get("hbm_bytes_per_launch")
                out["traffic_source"] = {"from_committed_profile": True, "file": "profiles/traffic_l1box.json",
                                         "collected": d.get("collected", time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(p)))),
                                         "how": d.get("how", "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/profile.sh)")}
        except Exception:
            pass
    return out


def _time_op(s, L, ctx, fn, iters=10, rounds=5):
    """median over `rounds` of (HIP-event time of `iters` back-to-back calls) / iters, after a clock-ramp round"""
    ms = ctypes.c_float()
    out = []
    for r in range(rounds + 1):
        s._lib.check(L.spx_timer_start(ctx))
        for _ in range(iters):
            fn()
        s._lib.check(L.spx_timer_stop(ctx, ctypes.byref(ms)))
        if r:
            out.append(ms.value / iters)
    out.sort()
    return out[len(out) // 2]


def _extra(s, L, ctx, dev, n, torch):
    """Secondary lines: the other BASELINE configs at full size (ms per call, G-elements/s, GB/s on the
    algorithmic byte count)."""
    res = {}

    def synth(m, stream, kind, scale=1.0):
        """spx_synth_fill(seed + 1000, stream, kind): the whole bench line is torch-RNG-free (kind 0: U(-1/2, 1/2), 1: ~N(0, 1))"""
        t = torch.empty(m, dtype=torch.float64, device=dev)
        s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), m, SYNTH_SEED + 1000, stream, kind, scale))
        return t

    xk, sj, q = synth(n, 0, 1), synth(n, 1, 0), synth(n, 2, 1)
    y = torch.empty_like(q)
    chi = s.NormLinf(1.0)

    def line(name, psi, bytes_per_elem, nel, yy, qq, kernel):
        # ms = avg_launch_ms: HIP events around 10 back-to-back calls on the launching stream / 10, median of 5 rounds.
        # kernel = the dominant kernel of the call as rocprofv3 --kernel-trace names it (profiles/r04_all_ops_kernel_stats.txt;
        # tools/r3/check_bench_kernels.py checks every string here against that file)
        ms = _time_op(s, L, ctx, lambda: s.prox_bang(yy, psi, qq, 1.0))
        res[name] = {"ms": round(ms, 4), "avg_launch_ms": round(ms, 4), "kernel": kernel, "gelem_s": round(nel / ms / 1e6, 2),
                     "gbs_algorithmic": round(bytes_per_elem * nel / ms / 1e6, 1),
                     "frac_of_peak": round(bytes_per_elem * nel / ms / 1e6 / HBM_PEAK_GBS, 4)}

    line("ShiftedNormL1", s.shifted(s.shifted(s.NormL1(1.0), xk), sj), 32, n, y, q, "k_sep_lds<OpL1, 6, false, false>")
    line("ShiftedNormL0", s.shifted(s.shifted(s.NormL0(1.0), xk), sj), 32, n, y, q, "k_sep_lds<OpL0, 6, false, false>")
    line("ShiftedNormL0Box", s.shifted(s.shifted(s.NormL0(1.0), xk, 1.0, chi), sj), 32, n, y, q, "k_sep_lds<OpL0Box, 6, false, false>")
    line("ShiftedRootNormLhalf", s.shifted(s.shifted(s.RootNormLhalf(1.0), xk), sj), 32, n, y, q, "k_sep_lds<OpLhalf, 6, false, false>")
    line("ShiftedRootNormLhalfBox", s.shifted(s.shifted(s.RootNormLhalf(1.0), xk, 1.0, chi), sj), 32, n, y, q, "k_sep_vec<OpLhalfBox, 4, false, false, true>")
    lv = -1.0 - 0.1 * (synth(n, 3, 0) + 0.5)
    uv = 1.0 + 0.1 * (synth(n, 4, 0) + 0.5)
    line("ShiftedNormL1Box_vector_bounds", s.shifted(s.shifted(s.NormL1(1.0), xk, lv, uv), sj), 48, n, y, q, "k_sep_lds<OpL1Box, 3, true, false>")
    del lv, uv
    # Float32 form of the headline operator (the reference is generic in R <: Real): 16 B/element, bit-exact in fp32
    x32, s32, q32 = xk.float(), sj.float(), q.float()
    y32 = torch.empty_like(q32)
    line("ShiftedNormL1Box_float32", s.shifted(s.shifted(s.NormL1(1.0), x32, 1.0, chi), s32), 16, n, y32, q32,
         "k_sep_f32<F32L1Box, false, false>")
    res["ShiftedNormL1Box_float32"]["dtype"] = "f32"
    del x32, s32, q32, y32
    # iprox!: g, d, xk, sj -> y, 40 B/element
    d = synth(n, 5, 0) + 1.0

    def iline(name, psi):
        ms = _time_op(s, L, ctx, lambda: s.iprox_bang(y, psi, q, d, check=False))
        res[name] = {"ms": round(ms, 4), "avg_launch_ms": round(ms, 4),
                     "kernel": "k_sep_lds<OpIproxL0Box, 4, false, false>" if "L0" in name else "k_sep_vec<OpIproxL1Box, 4, false, false, true>",
                     "gelem_s": round(n / ms / 1e6, 2), "gbs_algorithmic": round(40 * n / ms / 1e6, 1),
                     "frac_of_peak": round(40 * n / ms / 1e6 / HBM_PEAK_GBS, 4)}

    # psi(y): reduction over y, xk, sj: 24 B/element, returns a host double (synchronous)
    psi_l1b = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)
    s.prox_bang(y, psi_l1b, q, 1.0)  # y = the prox: the (feasible) point a solver evaluates psi at
    psi_l1b(y)  # untimed: the first call may grow the context scratch
    t0 = time.perf_counter()
    for _ in range(10):
        psi_l1b(y)
    ms = (time.perf_counter() - t0) / 10 * 1e3
    res["objective_ShiftedNormL1Box"] = {"ms": round(ms, 4), "gelem_s": round(n / ms / 1e6, 2),
                                         "gbs_algorithmic": round(24 * n / ms / 1e6, 1),
                                         "frac_of_peak": round(24 * n / ms / 1e6 / HBM_PEAK_GBS, 4),
                                         "kernel": "k_obj<double, TermL1, 1> (one launch: its last workgroup adds the partial sums)",
                                         "note": "host wall time per call incl. the read-back of the value"}
    # the same value left on the device (spx_ctx_set_value_target): no read-back, HIP-event time of back-to-back calls
    vout = torch.zeros(1, dtype=torch.float64, device=dev)
    with s.device_values(vout):
        ms = _time_op(s, L, ctx, lambda: psi_l1b(y))
    res["objective_ShiftedNormL1Box_device_value"] = {"ms": round(ms, 4), "avg_launch_ms": round(ms, 4),
                                                      "kernel": "k_obj<double, TermL1, 1> (one launch: its last workgroup adds the partial sums)", "gelem_s": round(n / ms / 1e6, 2),
                                                      "gbs_algorithmic": round(24 * n / ms / 1e6, 1),
                                                      "frac_of_peak": round(24 * n / ms / 1e6 / HBM_PEAK_GBS, 4),
                                                      "note": "value stored in a device double, nothing read back"}
    # prox! fused with h at the result (one pass instead of prox! + psi(y)); host wall time, the value is read back
    s.prox_value_bang(y, psi_l1b, q, 1.0)
    t0 = time.perf_counter()
    for _ in range(10):
        s.prox_value_bang(y, psi_l1b, q, 1.0)
    ms = (time.perf_counter() - t0) / 10 * 1e3
    res["prox_value_ShiftedNormL1Box"] = {"ms": round(ms, 4), "gelem_s": round(n / ms / 1e6, 2),
                                          "gbs_algorithmic": round(32 * n / ms / 1e6, 1),
                                          "frac_of_peak": round(32 * n / ms / 1e6 / HBM_PEAK_GBS, 4),
                                          "kernel": "k_sep_lds<WithValue<OpL1Box, HTermL1>, 6, false, false> (+ k_value_reduce)",
                                          "note": "prox! and h(xk + sj + y) in one pass (separately: the two lines above); "
                                                  "host wall time incl. the read-back of the value"}
    with s.device_values(vout):
        ms = _time_op(s, L, ctx, lambda: s.prox_value_bang(y, psi_l1b, q, 1.0))
    res["prox_value_ShiftedNormL1Box_device_value"] = {"ms": round(ms, 4), "avg_launch_ms": round(ms, 4),
                                                       "kernel": "k_sep_lds<WithValue<OpL1Box, HTermL1>, 6, false, false> (+ k_value_reduce)",
                                                       "gelem_s": round(n / ms / 1e6, 2), "gbs_algorithmic": round(32 * n / ms / 1e6, 1),
                                                       "frac_of_peak": round(32 * n / ms / 1e6 / HBM_PEAK_GBS, 4),
                                                       "note": "value stored in a device double, nothing read back"}
    iline("iprox_ShiftedNormL1Box", s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj))
    iline("iprox_ShiftedNormL0Box", s.shifted(s.shifted(s.NormL0(1.0), xk, 1.0, chi), sj))
    del d
    # ShiftedNormL1B2: reduction passes + scalar root find inside ONE launch (k_b2_coop), no host round trip
    psi_b2 = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj)
    s.prox_bang(y, psi_b2, q, 1.0)
    ms = _time_op(s, L, ctx, lambda: s.prox_bang(y, psi_b2, q, 1.0), iters=5, rounds=3)
    res["ShiftedNormL1B2"] = {"ms": round(ms, 4), "avg_launch_ms": round(ms, 4), "kernel": "k_b2_coop<false, 1, 1024, true, false>",
                              "gelem_s": round(n / ms / 1e6, 2), "gbs_algorithmic": round(32 * n / ms / 1e6, 1),
                              "frac_of_peak": round(32 * n / ms / 1e6 / HBM_PEAK_GBS, 4),
                              "note": "algorithmic 32 B/element; the call streams 24 + 32 = 56 B/element in ONE launch (round 3: the first "
                                      "pass classifies every element against a bracket around the sample's root -- fixed sums + ~1-2 % "
                                      "candidates -- the root is found on those, the second pass stores y; round 2: 80 B/element, 1.34 ms)"}
    # host-pointer form of the headline operator (spx_host_prox_l1_box): PCIe-inclusive, pageable numpy vectors
    nh = min(n, 10**7)
    hx, hs, hq = (t[:nh].cpu().numpy() for t in (xk, sj, q))
    psi_h = s.shifted(s.shifted(s.NormL1(1.0), hx, 1.0, chi), hs)
    s.prox(psi_h, hq, 1.0)
    t0 = time.perf_counter()
    for _ in range(3):
        s.prox(psi_h, hq, 1.0)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    res["host_form_ShiftedNormL1Box_pcie_inclusive"] = {
        "n": nh, "ms": round(ms, 3), "gelem_s": round(nh / ms / 1e6, 3), "gbs_over_pcie": round(32 * nh / ms / 1e6, 1),
        "note": "3 vectors H2D + y D2H per call from pageable host memory; never the headline value"}
    del hx, hs, hq, psi_h
    r = max(1, n // 100)
    # top-r is a sequence: k_s2_front, k_s2_main (dominant, ~88 % of the time), k_s2_scan_verify, k_s2_compact, k_s2_finish and
    # k_s2_tail, which returns at once on generic data; ms is the whole call
    TOPR = "k_s2_main<true, true> (+ k_s2_front<4>, k_s2_scan_verify, k_s2_compact<true, true>, k_s2_finish<true, true>, k_s2_tail<true>)"
    TOPR16 = TOPR.replace("k_s2_front<4>", "k_s2_front<16>")   # (a cut in the bulk of a vector of >= 2^26 elements: 16 samples per lane)
    line("ShiftedIndBallL0BInf_r=n/100", s.shifted(s.shifted(s.IndBallL0(r), xk, 1.0, chi), sj), 32, n, y, q, TOPR)
    # the same operator at the two ends of r (band without an upper end / widest band): tools/sweep_topr.py has the rest
    line("ShiftedIndBallL0BInf_r=1000", s.shifted(s.shifted(s.IndBallL0(min(1000, n)), xk, 1.0, chi), sj), 32, n, y, q, TOPR)
    line("ShiftedIndBallL0BInf_r=n/2", s.shifted(s.shifted(s.IndBallL0(max(1, n // 2)), xk, 1.0, chi), sj), 32, n, y, q,
         TOPR16 if n >= (1 << 26) else TOPR)
    # the same operator on TIE-HEAVY data (SURVEY 8d, config 3's tie-stress variant; q rounded to multiples of 1/4: the r-th
    # largest |v| is shared by ~1 % of the vector).  Round 3: the tied key is counted as a class in the same single pass, the
    # index cut comes from a prefix sum over the per-wave counts (k_s2_tail); xk and sj in buffers of their own (honest traffic)
    z1, z2 = torch.zeros_like(xk), torch.zeros_like(sj)
    q4 = torch.round(q * 4.0) / 4.0                             # (with xk = sj = 0 the lattice is exact)
    for rr, tag in ((r, "n/100"), (max(1, n // 2), "n/2")):
        psi_t = s.shifted(s.shifted(s.IndBallL0(rr), z1, 1.0, chi), z2)
        s.prox_bang(y, psi_t, q4, 1.0)
        ms = _time_op(s, L, ctx, lambda: s.prox_bang(y, psi_t, q4, 1.0), iters=5, rounds=3)
        res["ShiftedIndBallL0BInf_r=%s_ties" % tag] = {
            "ms": round(ms, 4), "avg_launch_ms": round(ms, 4), "kernel": TOPR16 if (tag == "n/2" and n >= (1 << 26)) else TOPR,
            "gelem_s": round(n / ms / 1e6, 2),
            "gbs_algorithmic": round(32 * n / ms / 1e6, 1), "frac_of_peak": round(32 * n / ms / 1e6 / HBM_PEAK_GBS, 4),
            "note": "xk = sj = 0, q on a 1/4 lattice: ties at the threshold; round 2: 2.6-5.8 ms (exact radix select, ~12 passes)"}
    del q4, psi_t, z1, z2
    # per-call latency at solver-iteration sizes: the two operators with a data-dependent scalar (r-th largest, trust-region
    # root) run as ONE launch with in-launch rendezvous, nothing read back (us per call, HIP events over 50 back-to-back calls)
    # (round 3: up to 2^22 elements the vector stays on chip -- v / xk parked in LDS above 2^20 / 2^21 elements, registers below)
    # (round 4: up to 6 Mi elements for top-r -- 16 elements per lane in LDS + 8 in registers, k_sel_lds<.., 8>)
    for nn in (6_000_000, 4_000_000, 1_000_000, 100_000, 10_000):
        if nn > n:
            continue
        xs, ss_, qs, ys = xk[:nn], sj[:nn], q[:nn], y[:nn]
        sel_kernel = ("k_sel_lds<true, true, double, 8>" if nn > (1 << 22) else "k_sel_lds<true, true, double, 0>" if nn > (1 << 20)
                      else "k_sel_coop<true, true, double>" if nn > 8192 else "k_sel_small<true, double>")
        for name, psi_s, kern in (
                ("ShiftedIndBallL0BInf_r=n/100_n=%d" % nn, s.shifted(s.shifted(s.IndBallL0(max(1, nn // 100)), xs, 1.0, chi), ss_), sel_kernel),
                ("ShiftedIndBallL0BInf_r=n/2_n=%d" % nn, s.shifted(s.shifted(s.IndBallL0(max(1, nn // 2)), xs, 1.0, chi), ss_), sel_kernel),
                ("ShiftedNormL1B2_n=%d" % nn, s.shifted(s.shifted(s.NormL1(1.0), xs, 1.0, s.NormL2(1.0)), ss_),
                 "k_b2_coop<false, 1, 1024, true, false>" if nn > (1 << 22) else "k_b2_coop<true, 16, 1024, true, true>" if nn > (1 << 21)
                 else "k_b2_coop<true, 16, 512, true, false>")):
            s.prox_bang(ys, psi_s, qs, 1.0)
            ms = _time_op(s, L, ctx, lambda: s.prox_bang(ys, psi_s, qs, 1.0), iters=50, rounds=5)
            res[name] = {"us": round(ms * 1e3, 2), "avg_launch_ms": round(ms, 5), "kernel": kern, "n": nn,
                         "note": "one launch per call; 32 B/element at 8 TB/s would be %.2f us" % (32 * nn / 8e6)}
    # ONE group over the whole vector = shifted(NormL2(lambda), xk), the reference's default GroupNormL2 (src/shiftedGroupNormL2.jl:34-35,
    # src/shiftedGroupNormL2Binf.jl:48-49), and a handful of ragged groups of 0.05 n .. 0.33 n: a team of workgroups per group
    # (csrc/spx_group_team.hip; round 3: one workgroup per group, 384 / 1349 ms at n = 1e8, profiles/r04_big_groups_baseline.txt).
    # 32 B/element algorithmic; the streamed form moves 24 + 32 = 56 B/element (one reducing pass, one storing pass).
    cuts = sorted(set([0, n] + [int(n * f) for f in (0.09, 0.22, 0.31, 0.55, 0.6, 0.93)]))
    layouts = (("1x%d" % n, s.GroupNormL2([0.5 * n ** 0.5])),
               ("7ragged_n=%d" % n, s.GroupNormL2([0.5 * (b - a) ** 0.5 for a, b in zip(cuts, cuts[1:])], [range(a, b) for a, b in zip(cuts, cuts[1:])])))
    big = n > 2_359_296  # (beyond 256 workgroups x 9216 elements the team form streams)
    for tag, hg in layouts:
        psi_g, psi_gb = s.shifted(s.shifted(hg, xk), sj), s.shifted(s.shifted(hg, xk, 1.0, chi), sj)
        line("ShiftedGroupNormL2_" + tag, psi_g, 32, n, y, q, "k_group_team<false, true, 1>" if big else "k_group_team<false, true, 0>")
        line("ShiftedGroupNormL2Binf_" + tag, psi_gb, 32, n, y, q,
             "k_group_team<true, true, 2> (+ k_group_team<true, true, 1>, which returns at once)" if big else "k_group_team<true, true, 0>")
        for nm, pg in (("objective_ShiftedGroupNormL2_" + tag, psi_g), ("objective_ShiftedGroupNormL2Binf_" + tag, psi_gb)):
            s.prox_bang(y, pg, q, 1.0)
            with s.device_values(vout):
                ms = _time_op(s, L, ctx, lambda: pg(y))
            res[nm] = {"ms": round(ms, 4), "avg_launch_ms": round(ms, 4), "kernel": "k_obj_chunks<%d> (+ k_obj_chunk_groups, k_obj_final)" % (2 if "Binf" in nm else 0),
                       "gelem_s": round(n / ms / 1e6, 2), "gbs_algorithmic": round(24 * n / ms / 1e6, 1),
                       "frac_of_peak": round(24 * n / ms / 1e6 / HBM_PEAK_GBS, 4),
                       "note": "psi(y) at the prox, 24 B/element, value stored in a device double"}
    for nn in (4_000_000, 1_000_000, 100_000):   # per-call latency of the one-group forms at solver-iteration sizes
        if nn > n:
            continue
        xs, ss_, qs, ys = xk[:nn], sj[:nn], q[:nn], y[:nn]
        hs_ = s.GroupNormL2([0.5 * nn ** 0.5])
        for name, psi_s, kern in (("ShiftedGroupNormL2_1x%d" % nn, s.shifted(s.shifted(hs_, xs), ss_), "k_group_team<false, true, %d>" % (1 if nn > 2_359_296 else 0)),
                                  ("ShiftedGroupNormL2Binf_1x%d" % nn, s.shifted(s.shifted(hs_, xs, 1.0, chi), ss_),
                                   "k_group_team<true, true, 2>" if nn > 2_359_296 else "k_group_team<true, true, 0>")):
            if name in res:
                continue
            s.prox_bang(ys, psi_s, qs, 1.0)
            ms = _time_op(s, L, ctx, lambda: s.prox_bang(ys, psi_s, qs, 1.0), iters=50, rounds=5)
            res[name] = {"us": round(ms * 1e3, 2), "avg_launch_ms": round(ms, 5), "kernel": kern, "n": nn,
                         "note": "one group over the vector; 32 B/element at 8 TB/s would be %.2f us" % (32 * nn / 8e6)}
    # group config: (n // 100) groups of 128  (10^6 x 128 at n = 10^8)
    ng = max(1, n // 100)
    m = ng * 128
    del xk, sj, q, y
    torch.cuda.empty_cache()
    xk, sj, q = synth(m, 6, 1), synth(m, 7, 0), synth(m, 8, 1)
    y = torch.empty_like(q)
    lam = synth(ng, 3, 0) + 1.0                                   # U(0.5, 1.5): SURVEY 8d C5
    h = s.GroupNormL2.uniform(lam, 128)
    bpe = 32 + 8 / 128
    line("ShiftedGroupNormL2_%dx128" % ng, s.shifted(s.shifted(h, xk), sj), bpe, m, y, q, "k_group_reg<16, 8, false, true, false, true>")
    line("ShiftedGroupNormL2Binf_%dx128" % ng, s.shifted(s.shifted(h, xk, 1.0, chi), sj), bpe, m, y, q, "k_group_reg<8, 16, true, true, false, true>")
    # small groups (round 3: register tiles of one or two lanes per group): n / 8 groups of 8 on the first n elements
    ng8 = max(1, min(m, n) // 8)
    m8 = ng8 * 8
    lam8 = synth(ng8, 4, 0) + 1.0
    h8 = s.GroupNormL2.uniform(lam8, 8)
    line("ShiftedGroupNormL2_%dx8" % ng8, s.shifted(s.shifted(h8, xk[:m8]), sj[:m8]), 32 + 1, m8, y[:m8], q[:m8], "k_group_reg<4, 4, false, true, false, false>")
    line("ShiftedGroupNormL2Binf_%dx8" % ng8, s.shifted(s.shifted(h8, xk[:m8], 1.0, chi), sj[:m8]), 32 + 1, m8, y[:m8], q[:m8], "k_group_reg<1, 8, true, true, false, true>")
    del lam8, h8
    # a sparse iterate under a strong lambda: 90 % of the groups of xk are zero, sigma*lambda above ||S|| for most groups
    # (the reversed-bracket regime of the reference; tools/sweep_params.py has the full sweep)
    keep = (synth(ng, 9, 0) < -0.4).to(torch.float64).repeat_interleave(128)
    xk.mul_(keep)
    del keep
    h30 = s.GroupNormL2.uniform(lam * 30.0, 128)
    line("ShiftedGroupNormL2Binf_%dx128_sparse_iterate" % ng, s.shifted(s.shifted(h30, xk, 1.0, chi), sj), bpe, m, y, q, "k_group_reg<8, 16, true, true, false, true>")
    return res


def _graph_iteration(s, dev, torch):
    """One solver-style iteration at solver-iteration sizes -- prox!(ShiftedNormL1Box), psi(y) left on the device,
    prox!(ShiftedIndBallL0BInf), prox!(ShiftedNormL1B2) -- issued call by call from Python and replayed as ONE captured graph
    (hipGraph through torch.cuda.CUDAGraph; libspx's graph-safe mode): wall time per iteration, 200 iterations, one
    synchronisation at the end."""
    res = {}
    side = torch.cuda.Stream(device=dev)
    for nn in (10_000, 1_000_000):
        with torch.cuda.stream(side):
            gen = torch.Generator(device=dev).manual_seed(7)
            xk = torch.randn(nn, dtype=torch.float64, device=dev, generator=gen)
            sj = torch.rand(nn, dtype=torch.float64, device=dev, generator=gen) - 0.5
            q = torch.randn(nn, dtype=torch.float64, device=dev, generator=gen)
            ys = [torch.empty_like(q) for _ in range(3)]
            val = torch.zeros(1, dtype=torch.float64, device=dev)
            chi = s.NormLinf(1.0)
            p_box = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)
            p_top = s.shifted(s.shifted(s.IndBallL0(max(1, nn // 100)), xk, 1.0, chi), sj)
            p_b2 = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj)

            def iteration():
                s.prox_bang(ys[0], p_box, q, 1.0)
                with s.device_values(val):
                    p_box(ys[0])
                s.prox_bang(ys[1], p_top, q, 1.0)
                s.prox_bang(ys[2], p_b2, q, 1.0)

            for _ in range(3):
                iteration()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                iteration()
            reps = 200
            for _ in range(5):
                g.replay()
            side.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            side.synchronize()
            t_graph = (time.perf_counter() - t0) / reps
            for _ in range(5):
                iteration()
            side.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                iteration()
            side.synchronize()
            t_eager = (time.perf_counter() - t0) / reps
        res["n=%d" % nn] = {"us_per_iteration_python_calls": round(t_eager * 1e6, 1), "us_per_iteration_graph_replay": round(t_graph * 1e6, 1),
                            "kernels": "4 calls, 4 kernels: k_sep_lds<OpL1Box>, k_obj<TermL1,1>, k_sel_coop / k_sel_small, k_b2_coop (+ the zero-fill nodes of the in-launch synchronised kernels in the graph)"}
    return res


def _cpu_baseline(s, psi, q, xk, sj, y, n, torch):
    """The CPU oracle = literal C restatement of src/shiftedNormL1Box.jl:89-125, one thread (the reference
    is single-threaded Julia), on a bounded sample of the same workload; also used as a parity spot check."""
    from oracle import oracle
    import numpy as np
    m = min(n, 100_000_000)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    xh, sh, qh = (oracle.synth_fill(m, SYNTH_SEED, stream, kind, 1.0, ncpu) for stream, kind in ((0, 1), (1, 0), (2, 1)))
    with np.errstate(over="ignore"):  # wrap-around sum of the bit patterns, host vs device: same inputs without a copy
        inputs_match = all(int(h.view(np.int64).sum()) == int(t[:m].view(torch.int64).sum())
                           for h, t in ((xh, xk), (sh, sj), (qh, q)))
    best = None
    reps = 0
    t_all = time.perf_counter()
    while reps < 3 and (time.perf_counter() - t_all) < 20.0:
        t0 = time.perf_counter()
        ref = oracle.prox_l1_box(qh, xh, sh, 1.0, 1.0, -1.0, 1.0)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
    s.prox_bang(y, psi, q, 1.0)
    same = bool(np.array_equal(y[:m].cpu().numpy().view(np.int64), ref.view(np.int64)))
    # all host cores on the same loop (OpenMP chunks): NOT the reference, which is single-threaded -- an upper bound for the CPU
    out_mt = np.empty_like(ref)
    best_mt = None
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.prox_l1_box_mt(qh, xh, sh, 1.0, 1.0, -1.0, 1.0, ncpu, out=out_mt)
        dt = time.perf_counter() - t0
        best_mt = dt if best_mt is None else min(best_mt, dt)
    all_cores = {"value": round(m / best_mt / 1e9, 4), "unit": "G-elements/s", "cores": ncpu, "kind": "port, OpenMP over all host cores",
                 "note": "not the reference (single-threaded Julia): an upper bound for this memory-bound loop on the host",
                 "same_bits_as_single_thread": bool(np.array_equal(out_mt.view(np.int64), ref.view(np.int64)))}
    return {"value": round(m / best / 1e9, 4), "unit": "G-elements/s", "cores": 1, "kind": "port", "all_cores": all_cores,
            "host_cores_available": os.cpu_count(),
            "sample": "first %d elements of the same workload, best of %d runs (%.2f s each)" % (m, reps, best),
            "note": "reference (Julia) cannot run in this image; port = oracle/spx_oracle.c, gcc -O2 -ffp-contract=off",
            "inputs_regenerated_on_host_match_device": inputs_match, "gpu_bit_exact_on_sample": same}


def _cpu_other_configs():
    """The single-thread CPU port (oracle) on bounded samples of the other BASELINE configs, for the GPU/CPU picture of
    each operator -- reported, never a target (the Julia reference itself cannot run here)."""
    from oracle import oracle
    import numpy as np
    rng = np.random.default_rng(20250613)

    def data(m):
        return rng.normal(size=m), rng.uniform(-0.5, 0.5, size=m), rng.normal(size=m)

    res = {}

    def timed(name, m, fn):
        dt = None
        for _ in range(2):  # best of two: the first run also pays the page faults of the result vector
            t0 = time.perf_counter()
            fn()
            d1 = time.perf_counter() - t0
            dt = d1 if dt is None else min(dt, d1)
        res[name] = {"gelem_s": round(m / dt / 1e9, 5), "sample_elements": m, "seconds": round(dt, 2), "cores": 1}

    m = 20_000_000
    x, sj, q = data(m)
    timed("ShiftedNormL0Box", m, lambda: oracle.prox_l0_box(q, x, sj, 1.0, 1.0, -1.0, 1.0))
    m = 5_000_000
    x, sj, q = x[:m], sj[:m], q[:m]
    timed("ShiftedRootNormLhalfBox", m, lambda: oracle.prox_lhalf_box(q, x, sj, 1.0, 1.0, -1.0, 1.0))
    timed("ShiftedIndBallL0BInf_r=n/100", m, lambda: oracle.prox_indball_l0_binf(q, x, sj, m // 100, 1.0))
    ng = 20_000
    m = ng * 128
    lam = rng.uniform(0.5, 1.5, size=ng)
    timed("ShiftedGroupNormL2Binf_%dx128" % ng, m, lambda: oracle.prox_group_l2_binf(q[:m], x[:m], sj[:m], lam, 1.0, 1.0, gsize=128))
    return res


if __name__ == "__main__":
    main()
