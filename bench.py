#!/usr/bin/env python3
"""Benchmark of the hot path: batched cold-start SQP-RTI solves on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  Prints ONE JSON line on rank 0.

A "step" = one batch of B independent NMPC instances solved start to finish (linearise ->
active-set / interior-point QP -> full SQP step) with the inputs already resident in HBM; when N>1
the first-stage commands u0 (the only exchange the path has) are moved by RCCL all-gathers, one per
group of --gather-every consecutive ticks, asynchronously to the following solves.
Workload = BASELINE.json configs[1]: B = 4096 near-hover initial states per GPU, horizon 20,
FP64, hover reference materialised per instance ([B,N,17]), x0 sample of seed 0 (config 2).
The other configs of the survey are flags: config 3 `--batch 65536 --dtype f32io --seed 1`, config 5 `--batch 1024 --horizon 600 --seed 5`
(tools/bench_table.sh runs them all).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
FP64_VEC_PEAK_TF = 78.6      # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FP32_VEC_PEAK_TF = 157.3


def algorithmic_bytes(N: int, esz: int, bcast: bool, traj_out: bool) -> int:
    """compulsory I/O of one solve."""
    n_in = 13 + (0 if bcast else N * 17 + 13)
    b = n_in * esz + 4 * esz + 4
    if traj_out:
        b += ((N + 1) * 13 + N * 4) * esz
    return b


def algorithmic_flops(N: int, n_kkt: float) -> float:
    """F = N*F_lin + n_kkt*N*F_kkt (dense-equivalent count, nominal).  n_kkt = KKT
    factorise-and-solve rounds: IPM iterations (1 factorisation + 2 solves) plus active-set passes
    (1 factorisation + 1 solve + 1 adjoint sweep), both priced at F_kkt = 11.9 kflop per stage."""
    return N * 7.7e3 + n_kkt * N * 11.9e3


def executed_flops(N: int, n_ipm: float, n_pass: float, shared: bool) -> float:
    """What the default path really executes (same per-stage prices): one linearisation when the cold
    start is shared; an active-set pass is one factorisation (8.27k) + one forward solve (1.8k); an
    IPM iteration is one factorisation + two solve pairs (11.9k).  Passes that resume from a Riccati
    checkpoint are priced as full passes (a slight over-count on <1 % of the instances)."""
    return (1 if shared else N) * 7.7e3 + N * (n_ipm * 11.9e3 + n_pass * 10.07e3)


def _physical_cores(allowed) -> int:
    """Distinct (package, core) pairs among the hardware threads this process may run on (SMT siblings count once)."""
    seen = set()
    try:
        for cpu in allowed:
            base = Path(f"/sys/devices/system/cpu/cpu{cpu}/topology")
            seen.add((int((base / "physical_package_id").read_text()), int((base / "core_id").read_text())))
    except Exception:
        return 0
    return len(seen)


def _cgroup_cpu_quota() -> float:
    """CPUs the container's cgroup grants (cpu.max / cfs quota), or 0.0 when there is no quota file or no limit."""
    try:
        f = Path("/sys/fs/cgroup/cpu.max")
        if f.exists():
            q, per = f.read_text().split()[:2]
            return 0.0 if q == "max" else float(q) / float(per)
        q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        per = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        return q / per if q > 0 else 0.0
    except Exception:
        return 0.0


def _best_thread_count(solve, make_x0, allowed: int):
    """The thread count that gives the highest rate on THIS box.  sched_getaffinity names every hardware thread of the host, but a GPU box
    gives one job a share of them (round 4: 256 threads visible, the rate of 256 OpenMP threads 5.6x one thread): a quota in the cgroup
    files is honoured if there is one, and the candidates (8, 16, 32, ... up to what is allowed) are measured on a short sample either way -
    `cores` of the baseline is then the number of threads actually used, as the contract asks."""
    quota = _cgroup_cpu_quota()
    top = allowed if quota <= 0 else max(1, min(allowed, int(round(quota))))
    cands = sorted({c for c in (4, 8, 16, 32, 64, 128, 256, top) if c <= top})
    x = make_x0(max(2048, 16 * top))
    best, table = cands[0], {}
    for c in cands:
        n = min(len(x), 128 * c)
        for _ in range(2):
            solve(x[:n], c)                                          # the OpenMP team of that size up and settled (the first regions after a resize are slow)
        table[c] = 0.0
        for _ in range(2):                                           # best of two
            t = time.perf_counter(); solve(x[:n], c); table[c] = max(table[c], n / (time.perf_counter() - t))
        if table[c] > table[best]:
            best = c
    return best, table, quota


def _timed_all_cores(solve, make_x0, cores: int, floor: int, target_s: float = 2.5, cap: int = 1 << 20):
    """Rate of `solve` over `cores` threads on a sample sized to ~target_s of wall time (never below `floor` = 64 x
    threads: every thread gets a run of instances long enough that scheduling and the first touch of its workspace do not
    show).  Pass 1 (floor-sized) warms the thread pool and calibrates, pass 2 warms the sized sample, pass 3 is timed."""
    x = make_x0(floor)
    t = time.perf_counter(); solve(x, cores); rate = floor / (time.perf_counter() - t)
    n = int(min(cap, max(floor, rate * target_s)))
    x = make_x0(n)
    solve(x, cores)
    t = time.perf_counter(); res = solve(x, cores); dt = time.perf_counter() - t
    return n / dt, n, res


def cpu_baseline(B_sample: int, N: int):
    """Two CPU restatements on this box's host cores, bounded samples of the config-2 instances (seed 0):
    (1) the oracle (oracle/nmpc_oracle.c): dense, unstructured, readable - the parity checker, same algorithm as the GPU default
        (active-set passes + interior point);
    (2) the structured host build of the lane kernels' own bodies (tests/hostsim: sparse model Jacobians, packed Riccati
        recursion on the 7 dense columns of A, no dense 13x13 loops) - plain interior point, the algorithm class HPIPM runs.
    Both are ports (`kind`), neither is acados: the reference's own CPU path cannot be built here.
    All-core figures: no allocation inside a solve (per-thread arenas / workspaces), static schedule, samples of >= 64 instances
    per hardware thread sized to ~2.5 s, two warm-up passes; `scaling_efficiency` = value / (threads x single-thread value)."""
    from oracle import oracle as O
    from rotors_mpc_controller_amd import _lib
    from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0
    c = O.default_config(N=N, qp_gamma=0.0, qp_polish=1)      # same algorithm as the GPU default
    yref, ye = O.hover_yref(c)
    x0 = sample_x0(B_sample, 0, **NEAR_HOVER)
    allowed = sorted(os.sched_getaffinity(0))
    phys = _physical_cores(allowed)
    out = O.solve_batch(c, x0, yref, ye, nthreads=min(len(allowed), 16))      # the parity sample
    mk = lambda n: sample_x0(n, 0, **NEAR_HOVER)
    osolve = lambda x, nt: O.solve_batch(c, x, yref, ye, nthreads=nt)
    cores, ctab, quota = _best_thread_count(osolve, mk, len(allowed))
    rate, n_all, _ = _timed_all_cores(osolve, mk, cores, max(64 * cores, 4096))
    n1 = 4096
    x1 = mk(n1)
    O.solve_batch(c, x1[:256], yref, ye, nthreads=1)
    t = time.perf_counter()
    O.solve_batch(c, x1, yref, ye, nthreads=1)
    r1 = n1 / (time.perf_counter() - t)
    share = (f"{len(allowed)} hardware threads visible ({phys} physical cores)" + (f", cgroup quota {quota:.1f} CPUs" if quota > 0 else ", no cgroup quota readable")
             + f"; thread count chosen by measurement {({k: round(v) for k, v in ctab.items()})}")
    row = dict(value=rate, unit="solves/s", cores=cores, visible_hardware_threads=len(allowed), physical_cores=phys or None, kind="port",
               sample=f"{n_all} near-hover instances (config-2 recipe, seed 0), OpenMP static schedule over instances, "
                      f"oracle/nmpc_oracle.c (dense restatement, not acados/HPIPM); third of three passes",
               single_thread_value=r1, scaling_efficiency=rate / (cores * r1),
               note=f"{share}; scaling_efficiency = value / (cores x single-thread value) with cores = the threads used")
    try:
        from tests import hostsim as H
        cfg = _lib.default_config(N=N, flags=_lib.FLAG_SHARE_COLD_START)
        yr, yre = hover_reference(N, cfg.mass * cfg.gravity / 4.0)
        hsolve = lambda x, nt: H.solve_batch(cfg, x, yr, yre, nthreads=nt)
        chk = hsolve(x0, min(len(allowed), 16))                          # the check: the oracle's own parity sample
        scores, stab, _ = _best_thread_count(hsolve, mk, len(allowed))
        rs, ns_all, hs = _timed_all_cores(hsolve, mk, scores, max(64 * scores, 4096))
        H.solve_batch(cfg, x1[:256], yr, yre, nthreads=1)
        t = time.perf_counter()
        H.solve_batch(cfg, x1, yr, yre, nthreads=1)
        rs1 = n1 / (time.perf_counter() - t)
        ok = (chk["status"] == 0) & (out["status"] == 0)
        row["structured"] = dict(value=rs, unit="solves/s", cores=scores, physical_cores=phys or None, kind="port", single_thread_value=rs1,
                                 scaling_efficiency=rs / (scores * rs1), thread_count_table={k: round(v) for k, v in stab.items()},
                                 sample=f"{ns_all} near-hover instances (config-2 recipe, seed 0), OpenMP static schedule over chunks of 8 lanes in "
                                        f"thread-private workspaces, host build of the lane kernel bodies nmpc_lane.hpp / nmpc_ipm.hpp "
                                        f"(tests/hostsim): structured Riccati, plain interior point; third of three passes",
                                 ipm_iterations_mean=float(hs["iters"].mean()),
                                 max_abs_u0_vs_oracle=float(np.abs(chk["u0"][ok] - out["u0"][ok]).max()),
                                 note="plain interior point to mu <= 1e-11 (no active-set shortcut: ~8x the factorisations of row 1 on this "
                                      "workload); the difference to the oracle's exact active-set answer is the interior point's own accuracy")
    except Exception as e:                                    # the baseline must never take the bench line down
        row["structured"] = dict(error=f"{type(e).__name__}: {e}")
    return out, row


def source_hash() -> str:
    """sha1 over the kernel sources (tools/source_hash.py): ties a committed PMC summary to the build it was captured on, and -
    compared with the hash compiled into the loaded binary (nmpc_version()) - shows a stale .so."""
    sys.path.insert(0, str(ROOT / "tools"))
    from source_hash import source_hash as sh
    return sh()


def relaunch_under_torchrun(n: int) -> None:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process group (before
    anything here has touched the GPU) and relay its output; the child ranks see WORLD_SIZE = N."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def timed_rate(solver, B, launch, steps, warmup, torch, dev):
    """solves/s and ms per step of `launch()` repeated `steps` times (device-resident inputs, one stream)."""
    for _ in range(warmup):
        launch()
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(steps):
        launch()
    torch.cuda.synchronize(dev)
    el = (time.perf_counter() - t) / steps
    return B / el, 1e3 * el


def secondary_rows(args, B, N, local, dev, tdt, npdt, torch, _lib, NmpcOcpSolver, x0_h, yref, yref_e, bcast):
    """The rows SURVEY 8(d) wants next to the headline ("report with and without" the shared cold start; the
    plain interior-point path, which is what the reference's HPIPM does; the aggressive initial-state set):
    same batch, same inputs unless stated, measured after the timed region."""
    from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, sample_x0
    rows = {}
    x0_aggr = torch.from_numpy(sample_x0(B, 0, **AGGRESSIVE).astype(npdt)).to(dev)
    x0_near = torch.from_numpy(x0_h.astype(npdt)).to(dev)
    u0 = torch.zeros(B, 4, dtype=tdt, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    variants = [("no_share", dict(flags=_lib.FLAG_TEAM_MAPPING), x0_near, "per-stage linearisation (what every warm-started tick runs)"),
                ("plain_ipm", dict(qp_polish=0), x0_near, "qp_polish = 0: Mehrotra interior point only, no active-set passes"),
                ("aggressive", dict(), x0_aggr, "aggressive x0 set (SURVEY 8d), seed 0: ~28 % of the instances hit a bound")]
    for name, over, x0, what in variants:
        cfg = _lib.default_config(N=N, max_batch=B, device=local,
                                  dtype=dict(f64=_lib.DTYPE_F64, f32=_lib.DTYPE_F32, f32io=_lib.DTYPE_F32IO)[args.dtype])
        cfg.update(**over)
        sv = NmpcOcpSolver(cfg)
        sv.set_timing(False)

        def launch(sv=sv, x0=x0):
            sv.solve_batch_device(B, x0.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), bcast, u0.data_ptr(),
                                  status_ptr=status.data_ptr(), stream=stream.cuda_stream)
        rate, ms = timed_rate(sv, B, launch, max(10, args.steps // 4), 5, torch, dev)
        sv.set_timing(True)
        launch()
        st = sv.stats()
        rows[name] = dict(value=rate, unit="solves/s", ms_per_step=ms, what=what,
                          ipm_iterations_mean=st["iter_mean"], ipm_iterations_max=st["iter_max"],
                          active_set_passes_mean=st["polish_mean"], active_set_passes_max=st["polish_max"],
                          status_histogram=st["n_status"])
        sv.close()
    # independent batches (Monte-Carlo use): two solver handles on two streams, batches alternate between them.  The
    # tail of one batch - the few waves whose instances need more active-set passes - runs beside the start of the next
    if args.dtype in ("f64", "f32io"):
        from rotors_mpc_controller_amd.pipeline import BatchPipeline
        pipe = BatchPipeline(_lib.default_config(N=N, max_batch=B, device=local,
                                                 dtype=dict(f64=_lib.DTYPE_F64, f32io=_lib.DTYPE_F32IO)[args.dtype]), depth=2)
        outs = [(torch.zeros(B, 4, dtype=tdt, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(2)]
        tick = [0]

        def launch2():
            u_i, s_i = outs[tick[0] & 1]
            tick[0] += 1
            pipe.submit(B, x0_near.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), bcast, u_i.data_ptr(), status_ptr=s_i.data_ptr(),
                        after_current_stream=False)          # inputs are resident long before
        rate, ms = timed_rate(None, B, launch2, 2 * max(10, args.steps // 4), 6, torch, dev)
        same = bool(torch.equal(outs[0][0], outs[1][0]) and int(outs[0][1].abs().sum()) == 0)
        rows["two_batches_in_flight"] = dict(value=rate, unit="solves/s", ms_per_step=ms, outputs_equal=same,
                                             what="independent batches alternate between two solver handles on two streams "
                                                  "(rotors_mpc_controller_amd.pipeline.BatchPipeline; the headline value is one "
                                                  "handle, one stream, launches back to back)")
        pipe.close()
    return rows


def latency_config1(local, _lib):
    """BASELINE.json configs[0] / BASELINE.md run C0: ONE instance, hover setpoint, the way the reference uses
    its solver (60 Hz, config/params.yaml:48).  Median wall time of (a) bare nmpc_solve() on the
    single-instance slot (linearisation point and references already set), (b) the whole
    PositionNMPC.solve(state, reference) facade call (64 set + solve + 42 get equivalents, warm-started)."""
    import statistics
    from rotors_mpc_controller_amd.controller import PositionNMPC
    from rotors_mpc_controller_amd.params import load_params
    from rotors_mpc_controller_amd.reference import ReferenceGenerator
    params = load_params()
    ctl = PositionNMPC(params, max_batch=1, device=local)
    gen = ReferenceGenerator(params["reference"])
    gen.update_hover_thrust(ctl.hover_thrust)
    ref = gen.build_horizon(ctl.horizon, ctl.dt)
    state = dict(position=np.array([0.0, 0.0, 0.5]), velocity=np.zeros(3), quaternion=np.array([1.0, 0, 0, 0]), body_rates=np.zeros(3))
    sv = ctl._solver
    sv.set_timing(False)
    for _ in range(20):
        ctl.solve(state, ref)
    tf, tb, ts = [], [], []
    for _ in range(200):
        t = time.perf_counter(); u, stt = ctl.solve(state, ref); tf.append(time.perf_counter() - t)
        assert stt == 0
    ctl._one_call = False            # the reference's own call sequence: 64 set + solve + 42 get per tick
    for _ in range(20):
        ctl.solve(state, ref)
    for _ in range(200):
        t = time.perf_counter(); u, stt = ctl.solve(state, ref); ts.append(time.perf_counter() - t)
        assert stt == 0
    for _ in range(200):
        t = time.perf_counter(); rc = sv.solve(); tb.append(time.perf_counter() - t)
        assert rc == 0
    return dict(bare_nmpc_solve=1e6 * statistics.median(tb), facade_PositionNMPC_solve=1e6 * statistics.median(tf),
                facade_reference_call_sequence=1e6 * statistics.median(ts),
                samples=200, statistic="median", workload="B = 1, hover setpoint, x0 = (0,0,0.5), N = 20, FP64, warm start",
                note="host wall time incl. two pinned-memory copies and one stream synchronisation per solve")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--seed", type=int, default=None, help="seed of the x0 sample (default: 0 on one GPU, 100 + rank on several - SURVEY 8d)")
    ap.add_argument("--dtype", choices=["f64", "f32", "f32io"], default="f64",
                    help="f32io (= f32 since round 5): FP32 device buffers, FP64 arithmetic (NMPC_DTYPE_F32IO); the JSON line's dtype "
                         "stays f64 - the arithmetic - with config.device_buffers = f32")
    ap.add_argument("--dist", choices=["near_hover", "aggressive"], default="near_hover")
    ap.add_argument("--yref", choices=["per_instance", "broadcast"], default="per_instance")
    ap.add_argument("--no-share", action="store_true", help="do not exploit the shared cold-start linearisation")
    ap.add_argument("--traj-out", action="store_true", help="also write the full x/u trajectories")
    ap.add_argument("--mapping", choices=["team", "lane"], default="team",
                    help="QP phase: 16 lanes per instance (team) or one instance per lane")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-solve-events", action="store_true",
                    help="keep the library's own HIP events around every solve inside the timed region")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="multi-GPU: ticks whose u0 share one RCCL all-gather (1 = a collective per tick, the default; "
                         "8 = the batched exchange, reported as a secondary row when N > 1)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary rows and the config-1 latency")
    ap.add_argument("--no-polish", action="store_true", help="plain interior point iteration (qp_polish = 0)")
    ap.add_argument("--condensed", action="store_true", help="partial-condensing kernel (NMPC_FLAG_CONDENSED_QP)")
    ap.add_argument("--polish-ckpt", type=int, default=None, help="override nmpc_config.qp_polish_ckpt")
    ap.add_argument("--polish-passes", type=int, default=None, help="override nmpc_config.qp_polish_passes (passes per attempt)")
    ap.add_argument("--polish-budget", type=int, default=None, help="override nmpc_config.qp_polish_budget (passes in total)")
    ap.add_argument("--cpu-sample", type=int, default=4096)
    args = ap.parse_args()
    if args.dtype == "f32":
        args.dtype = "f32io"      # BASELINE config 3 ("FP32"): served by FP32 buffers on the FP64 kernels

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # multi-process GPU work on this pool needs dmabuf IPC (already exported on the boxes)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_env:
        if world_env == 1 and args.gpus > 1:
            relaunch_under_torchrun(args.gpus)            # never returns
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world_env}")

    import torch
    import torch.distributed as dist

    from rotors_mpc_controller_amd import _lib
    from rotors_mpc_controller_amd.solver import NmpcOcpSolver
    from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solver has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("NMPC_BENCH_FORCE_DIST") == "1"   # the env knob rehearses the RCCL path on 1 GPU
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints a version banner on STDOUT when its communicator comes up - inside init_process_group when a device_id is given (eager
        # initialisation; found late in round 5: the banner stood in front of the JSON line of a one-rank rehearsal), at the first collective
        # otherwise: keep this process's stdout for the ONE JSON line of the contract - C-level writes go to stderr until the communicator exists
        sys.stdout.flush()
        _saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
            _w = torch.zeros(1, device=dev)
            dist.all_reduce(_w)
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(_saved_stdout, 1)
            os.close(_saved_stdout)

    B, N = args.batch, args.horizon
    tdt, npdt, esz = (torch.float64, np.float64, 8) if args.dtype == "f64" else (torch.float32, np.float32, 4)
    lib_dtype = dict(f64=_lib.DTYPE_F64, f32=_lib.DTYPE_F32, f32io=_lib.DTYPE_F32IO)[args.dtype]
    cfg = _lib.default_config(N=N, max_batch=B, device=local,
                              dtype=lib_dtype,
                              flags=(0 if args.no_share else _lib.FLAG_SHARE_COLD_START)
                              | (_lib.FLAG_TEAM_MAPPING if args.mapping == "team" else 0))
    if args.polish_ckpt is not None:
        cfg.update(qp_polish_ckpt=args.polish_ckpt)
    if args.polish_passes is not None:
        cfg.update(qp_polish_passes=args.polish_passes)
    if args.polish_budget is not None:
        cfg.update(qp_polish_budget=args.polish_budget)
    if args.no_polish:
        cfg.update(qp_polish=0)
    if args.condensed:
        cfg.update(flags=cfg.flags | _lib.FLAG_CONDENSED_QP)
    solver = NmpcOcpSolver(cfg)
    hover = cfg.mass * cfg.gravity / 4.0
    # seed 0 = config 2 (the headline), 100 + rank = config 4 (multi-GPU); --seed 1 / 5 give the samples of configs 3 / 5
    seed = (args.seed + (rank if world > 1 else 0)) if args.seed is not None else (0 if world == 1 else 100 + rank)   # (shards stay distinct)
    dist_kw = NEAR_HOVER if args.dist == "near_hover" else AGGRESSIVE
    x0_h = sample_x0(B, seed, **dist_kw)
    yref_h, yref_e_h = hover_reference(N, hover)
    bcast = args.yref == "broadcast"
    x0 = torch.from_numpy(x0_h.astype(npdt)).to(dev)
    if bcast:
        yref = torch.from_numpy(yref_h.astype(npdt)).to(dev)
        yref_e = torch.from_numpy(yref_e_h.astype(npdt)).to(dev)
    else:
        yref = torch.from_numpy(np.tile(yref_h, (B, 1, 1)).astype(npdt)).to(dev).contiguous()
        yref_e = torch.from_numpy(np.tile(yref_e_h, (B, 1)).astype(npdt)).to(dev).contiguous()
    # ---- the exchange of the commands (the only collective the path has: u0 [B,4] per rank and tick, 128 KiB in FP64).
    # Default (--gather-every 1, one collective per tick - the honest closed-loop exchange): the solve writes its commands
    # STRAIGHT into this rank's slot of the gather buffer [world][B][4] (in-place all-gather: no staging copy), and solve +
    # all-gather are ONE stream's work, captured once into a HIP graph and replayed per tick - no cross-stream event between a
    # tick's solve and its collective or between consecutive ticks (in round 2 those events broke the back-to-back dispatch of
    # the solves and cost 30 % of the rate in the one-rank rehearsal).  If the RCCL build cannot be captured the same two
    # operations are enqueued eagerly on the one stream (`exchange` in the JSON line says which ran).
    # --gather-every G > 1: the batched variant of round 1/2 - the u0 of G consecutive ticks fill one group buffer that one
    # ASYNCHRONOUS all-gather moves while the next group is solved into the other buffer.
    G = max(1, args.gather_every)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    xo = torch.zeros(B, N + 1, 13, dtype=tdt, device=dev) if args.traj_out else None
    uo = torch.zeros(B, N, 4, dtype=tdt, device=dev) if args.traj_out else None
    stream = torch.cuda.current_stream(dev)
    tick = [0]
    last = [0, 0]                                                # (group buffer, slot) of the latest solve
    exchange = "none"
    per_tick = use_dist and G == 1
    if per_tick:
        # two gather buffers, ticks alternate: while the collective of tick t moves buffer t % 2, the solve of tick t + 1 fills the
        # other one.  Inside the graph that is a second capture stream (graph edges, no run-time events): the small RCCL kernel runs
        # in the tail of the next solve, whose last waves leave most SIMDs idle anyway.
        gat2 = [torch.zeros(world, B, 4, dtype=tdt, device=dev) for _ in range(2)]
        u0g = [g_[rank].unsqueeze(0) for g_ in gat2]             # [1][B][4] views: this rank's slot of each gather buffer
        gathered = [g_.unsqueeze(1) for g_ in gat2]              # [world][1][B][4]
        pending = [None]
        KT = max(2, 2 * (int(os.environ.get("NMPC_BENCH_GRAPH_TICKS", "8")) // 2))      # ticks per graph replay (even)

        def solve_into(b, cuda_stream):
            solver.solve_batch_device(B, x0.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), bcast, gat2[b][rank].data_ptr(),
                                      status_ptr=status.data_ptr(),
                                      x_out_ptr=xo.data_ptr() if xo is not None else 0,
                                      u_out_ptr=uo.data_ptr() if uo is not None else 0, stream=cuda_stream)

        def gather(b):
            dist.all_gather_into_tensor(gat2[b].view(world * B, 4), gat2[b][rank])     # in place, on the current stream

        def tick_ops(b, cuda_stream):
            solve_into(b, cuda_stream)
            gather(b)

        solver.set_timing(False)
        for b in (0, 1):
            tick_ops(b, stream.cuda_stream)                      # communicator set-up, code objects: before any capture
        torch.cuda.synchronize(dev)
        graph, graph_ticks = None, 1
        mode = os.environ.get("NMPC_BENCH_EXCHANGE", "pipelined")       # pipelined | serial | eager
        if mode == "pipelined":
            try:
                sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
                sa.wait_stream(stream)
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, stream=sa):
                    ev_g = [None, None]                          # completion of the latest gather of each buffer
                    for t in range(KT):
                        b = t & 1
                        if ev_g[b] is not None:
                            sa.wait_event(ev_g[b])               # buffer b is free again
                        solve_into(b, sa.cuda_stream)
                        ev_s = torch.cuda.Event()
                        ev_s.record(sa)
                        sb.wait_event(ev_s)
                        with torch.cuda.stream(sb):
                            gather(b)
                            ev_g[b] = torch.cuda.Event()
                            ev_g[b].record(sb)
                    sa.wait_stream(sb)                           # join: the replay ends when every gather has landed
                torch.cuda.synchronize(dev)
                g_.replay()
                torch.cuda.synchronize(dev)
                graph, graph_ticks = g_, KT
                exchange = (f"in-place all-gather per tick, {KT} ticks per HIP-graph replay: the collective of tick t beside the solve of "
                            f"tick t + 1 (graph edges between two capture streams)")
            except Exception as e:
                exchange = f"(pipelined capture failed: {type(e).__name__}) "
                torch.cuda.synchronize(dev)
                mode = "serial"
        if graph is None and mode == "serial":
            try:
                gs = torch.cuda.Stream(dev)
                gs.wait_stream(stream)
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, stream=gs):
                    tick_ops(0, torch.cuda.current_stream(dev).cuda_stream)
                torch.cuda.synchronize(dev)
                g_.replay()
                torch.cuda.synchronize(dev)
                graph, graph_ticks = g_, 1
                exchange = (exchange if exchange != "none" else "") + "in-place all-gather per tick, solve + collective replayed from one HIP graph"
            except Exception as e:                               # RCCL not capturable on this build: same operations, eager
                exchange = f"in-place all-gather per tick, eager on one stream (graph capture failed: {type(e).__name__})"
                torch.cuda.synchronize(dev)
        if graph is None and not exchange.startswith("in-place"):
            exchange = "in-place all-gather per tick, eager on one stream"
        carry = [0]

        def step():
            # one tick; a replay of the pipelined graph covers graph_ticks of them (the timed step counts are multiples of it:
            # a remainder is run as single eager ticks)
            tick[0] += 1
            if graph is not None and graph_ticks > 1:
                carry[0] += 1
                if carry[0] == graph_ticks:
                    carry[0] = 0
                    graph.replay()
            elif graph is not None:
                graph.replay()
            else:
                tick_ops(0, stream.cuda_stream)

        def eager_step():
            tick_ops(0, stream.cuda_stream)

        def drain():
            # ticks counted but not yet replayed (steps % graph_ticks of them): run eagerly.  Called BEFORE the closing event of
            # the timed region, so that device_ms_per_step and roofline.frac of a multi-rank line cover every tick (round 3 ran them
            # inside fence(), after the event: 4 of the driver's 20 steps were missing from the device time of an N > 1 line)
            for _ in range(carry[0]):
                tick_ops(0, stream.cuda_stream)
            carry[0] = 0

        def fence():
            drain()
            dist.barrier()
            torch.cuda.synchronize(dev)
    else:
        def drain():
            pass

        u0g = [torch.zeros(G, B, 4, dtype=tdt, device=dev) for _ in range(2)]
        gathered = [torch.zeros(world, G, B, 4, dtype=tdt, device=dev) for _ in range(2)] if use_dist else None
        pending = [None, None]
        if use_dist:
            exchange = f"asynchronous all-gather of {G} ticks' commands (RCCL's own stream)"

        def flush(k):
            pending[k] = dist.all_gather_into_tensor(gathered[k].view(world * G * B, 4), u0g[k].view(G * B, 4), async_op=True)

        def step():
            t = tick[0]
            tick[0] += 1
            k, slot = (t // G) & 1, t % G
            if use_dist and slot == 0 and pending[k] is not None:
                pending[k].wait()                                   # group k's previous gather must have drained
                pending[k] = None
            last[0], last[1] = k, slot
            solver.solve_batch_device(B, x0.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), bcast, u0g[k][slot].data_ptr(),
                                      status_ptr=status.data_ptr(),
                                      x_out_ptr=xo.data_ptr() if xo is not None else 0,
                                      u_out_ptr=uo.data_ptr() if uo is not None else 0,
                                      stream=stream.cuda_stream)
            if use_dist and slot == G - 1:                          # RCCL over xGMI, asynchronous to the next group
                flush(k)

        eager_step = step

        def fence():
            if use_dist:
                t = tick[0]
                if t % G != 0:                                      # a partly filled group: gather it as well
                    k = (t // G) & 1
                    flush(k)
                    tick[0] = (t // G + 1) * G
                for k in (0, 1):
                    if pending[k] is not None:
                        pending[k].wait()
                        pending[k] = None
                dist.barrier()
            torch.cuda.synchronize(dev)

    # the timed region carries bench.py's own two HIP events only; the library's per-solve events
    # (two more stream operations per step) are switched back on for one untimed step afterwards,
    # which yields the isolated kernel times and the iteration statistics
    solver.set_timing(args.per_solve_events)
    # NMPC_BENCH_PREROLL=<n> (experiments, default 0): n untimed steps before the contract's W warmup steps - shows how much of a short run's
    # step time is the GPU still ramping its clocks; the driver's runs use the default
    preroll = int(os.environ.get("NMPC_BENCH_PREROLL", "0"))
    for _ in range(preroll):
        step()
    if preroll:
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    drain()                                                      # every tick of the timed region is enqueued before the closing event
    ev1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    solver.set_timing(True)
    eager_step()
    fence()
    st = solver.stats()                                          # HIP events of one extra, untimed (eagerly launched) step
    u0_h = u0g[last[0]][last[1]].cpu().numpy().astype(np.float64)
    if use_dist:                                                 # the gathered block of this rank is its own u0
        g = gathered[last[0]][rank, last[1]].cpu().numpy().astype(np.float64)
        assert np.array_equal(g, u0_h), "all-gather returned a different u0 block"
    status_h = status.cpu().numpy()

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        rate = world * B / (elapsed / args.steps)
        n_ipm = st["iter_mean"]
        # kernel duration = average over the timed region (HIP events on the launch stream, one fused
        # kernel per step); the isolated single-launch time of the extra step is reported beside it
        fused = st["ms_prepare"] == 0.0
        kern_s = (dev_ms if fused else st["ms_solve"]) * 1e-3
        alg_b = algorithmic_bytes(N, esz, bcast, args.traj_out)
        n_kkt = n_ipm + st["polish_mean"]
        flops = algorithmic_flops(N, n_kkt)
        f_peak = FP64_VEC_PEAK_TF
        hbm_alg_gbs = alg_b * B / kern_s / 1e9
        alu_tf = flops * B / kern_s / 1e12
        flops_x = executed_flops(N, n_ipm, st["polish_mean"], not args.no_share)
        alu_x_tf = flops_x * B / kern_s / 1e12
        # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of the SAME command
        # (tools/rocprof_capture.sh); FETCH_SIZE/WRITE_SIZE are KiB, FETCH_SIZE doubled per the
        # gfx950 note in MI355X_MICROARCH.md.  Only quoted for the configuration it was taken on.
        traffic, traffic_note = None, None
        sched = solver.last_schedule()
        split = sched["split"]                         # default FP64 path: first attempt by k_team_as
        inplace = sched["inplace"]                     # ... which also continues the attempts that fail (no work-list launch)
        kname = ("k_team_as" if split else "k_team_qp") if args.mapping == "team" else "k_ipm"
        pmc_file = ROOT / "profiles" / f"latest_{args.mapping}_b{B}_{args.dtype}_pmc_summary.json"
        if pmc_file.exists() and not args.no_share and not bcast and not args.traj_out and N == 20:
            pmc_all = json.loads(pmc_file.read_text())
            pmc = pmc_all.get(kname, {})
            # quoted only when the summary was captured on THIS build of the kernels (tools/summarize_pmc.py
            # records the hash of the kernel sources); otherwise null + a note, never a stale figure
            if _lib.library_source_hash() != source_hash():
                traffic_note = (f"stale binary: librotors_nmpc_hip.so was built from kernel sources {_lib.library_source_hash()}, "
                                f"the tree holds {source_hash()}")
            elif pmc_all.get("source_hash") != source_hash():
                traffic_note = f"stale profile: {pmc_file.name} was captured on kernel sources {pmc_all.get('source_hash')}, this build is {source_hash()}"
            elif "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                traffic = (2.0 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024.0
                tail = pmc_all.get("k_team_qp_list", {})
                if split and not inplace and "FETCH_SIZE" in tail and "WRITE_SIZE" in tail:      # both launches of a step
                    traffic += (2.0 * tail["FETCH_SIZE"]["mean"] + tail["WRITE_SIZE"]["mean"]) * 1024.0
        kernel_name = (("k_team_as" if inplace else "k_team_as + k_team_qp_list") if split else kname) if args.mapping == "team" else "k_ipm"
        hbm = dict(achieved=hbm_alg_gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm_alg_gbs / HBM_PEAK_GBS,
                   algorithmic_bytes_per_solve=alg_b,
                   measured_traffic_gbs=(traffic / kern_s / 1e9 if traffic is not None else None))
        # flops: `executed` prices what the path really does per solve (executed_flops above); `nominal` is the
        # dense-equivalent count of SURVEY 8d with every KKT round priced as a full IPM iteration and N
        # linearisations - kept for comparison, it over-counts the default path by ~1.9x
        flop = dict(achieved=alu_x_tf, peak=f_peak, unit="TFLOP/s", frac=alu_x_tf / f_peak,
                    flops_per_solve=flops_x, n_ipm_mean=n_ipm, n_passes_mean=st["polish_mean"], n_kkt_rounds_mean=n_kkt,
                    nominal=dict(achieved=alu_tf, frac=alu_tf / f_peak, flops_per_solve=flops))
        # Which roof: by compulsory traffic the path is arithmetic bound (~130 flop/B against a machine
        # balance of ~10).  The FP64 team kernel runs its factor and solve sweeps on v_mfma_f64_4x4x4 (the dense
        # FP64 matrix peak of gfx950 equals the vector peak, 78.6 TFLOP/s), so that is the roof quoted for it;
        # every other variant is a vector-ALU kernel and keeps the HBM line of the contract, with the flop
        # figures beside it.  Both sub-objects are always present.
        mfma_path = args.mapping == "team" and not args.condensed
        common = dict(kernel=kernel_name, traffic=traffic, traffic_source=(pmc_file.name if traffic is not None else traffic_note),
                      kernel_ms=kern_s * 1e3, kernel_ms_isolated=st["ms_solve"] + (0.0 if (split and inplace) else st.get("ms_tail", 0.0)), prepare_ms=st["ms_prepare"],
                      launches=(dict(k_team_as_ms_isolated=st["ms_solve"], k_team_qp_list_ms_isolated=(None if inplace else st["ms_tail"]),
                                     instances_in_second_launch=st["n_tail"],
                                     note=("kernel_ms = device time of one step = the one launch: k_team_as continues the first attempts that fail on the wave that "
                                           "made them (no work-list launch; instances_in_second_launch counts those continued)" if inplace
                                           else "kernel_ms = device time of one step = both launches (HIP events around the timed region)" if not sched["tail"]
                                           else "long horizon: k_team_as = preparation + first pass; the second figure is every later launch (block-parallel tail: "
                                                "k_block_sweep_tail / k_block_scan_tail / k_team_tail per step, then k_team_qp_list on the fallback list); "
                                                "instances_in_second_launch counts the instances that took an interior-point iteration"))
                                if split else None), hbm=hbm, alu=flop)
        if mfma_path:
            roof = dict(bound="mfma", achieved=alu_x_tf, peak=f_peak, unit="TFLOP/s", frac=alu_x_tf / f_peak, **common)
        else:
            roof = dict(bound="hbm", achieved=hbm_alg_gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm_alg_gbs / HBM_PEAK_GBS, **common)
        line = dict(metric=f"NMPC SQP-RTI solves/sec (N={N}, nx=13, nu=4) at batch={B} per GPU",
                    value=rate, unit="solves/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=ms_step, device_ms_per_step=dev_ms, higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype=("f64" if args.dtype == "f32io" else args.dtype), data="synthetic",
                    config=dict(workload=f"batch={B} random x0 around hover ({args.dist}, seed {seed}), N={N}, "
                                         f"{args.dtype.upper()}, cold start, hover yref {args.yref}",
                                batch_per_gpu=B, horizon=N, share_cold_start=not args.no_share, mapping=args.mapping,
                                device_buffers=("f32" if args.dtype != "f64" else "f64"),
                                traj_out=args.traj_out, preroll_steps=preroll, parallelism=f"batch-sharded x{world}, all-gather of u0 every {G} ticks",
                                exchange=exchange),
                    ipm_iterations=dict(mean=st["iter_mean"], min=st["iter_min"], max=st["iter_max"]),
                    active_set_passes=dict(mean=st["polish_mean"], max=st["polish_max"], accepted=st["n_polished"]),
                    status_histogram=st["n_status"], roofline=roof)
        if world == 1 and not args.no_cpu_baseline and args.dtype == "f64" and args.dist == "near_hover":
            ns = min(args.cpu_sample, B)
            ref, cb = cpu_baseline(ns, N)
            line["cpu_baseline"] = cb
            ok = (status_h[:ns] == 0) & (ref["status"] == 0)
            line["max_abs_u0_vs_oracle"] = float(np.abs(u0_h[:ns][ok] - ref["u0"][ok]).max())
            line["parity_note"] = "vs build CPU oracle; acados parity unpinned (SURVEY 8c)"
        line["source_hash"] = source_hash()
        line["binary_source_hash"] = _lib.library_source_hash()
        line["library_version"] = _lib.load().nmpc_version().decode()
    # ---- rows measured AFTER the timed region (never part of `value`)
    if use_dist and G == 1 and not args.no_secondary:
        # the batched exchange: u0 of 8 consecutive ticks share one asynchronous all-gather
        G8 = 8
        u0g8 = [torch.zeros(G8, B, 4, dtype=tdt, device=dev) for _ in range(2)]
        gat8 = [torch.zeros(world, G8, B, 4, dtype=tdt, device=dev) for _ in range(2)]
        pend8 = [None, None]
        solver.set_timing(False)

        def run8(n):
            for t in range(n):
                k, slot = (t // G8) & 1, t % G8
                if slot == 0 and pend8[k] is not None:
                    pend8[k].wait(); pend8[k] = None
                solver.solve_batch_device(B, x0.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), bcast, u0g8[k][slot].data_ptr(),
                                          status_ptr=status.data_ptr(), stream=stream.cuda_stream)
                if slot == G8 - 1:
                    pend8[k] = dist.all_gather_into_tensor(gat8[k].view(world * G8 * B, 4), u0g8[k].view(G8 * B, 4), async_op=True)
            for k in (0, 1):
                if pend8[k] is not None:
                    pend8[k].wait(); pend8[k] = None
            dist.barrier()
            torch.cuda.synchronize(dev)
        run8(2 * G8)
        n8 = max(2 * G8, (args.steps // (2 * G8)) * 2 * G8)
        t8 = time.perf_counter()
        run8(n8)
        e8 = torch.tensor([time.perf_counter() - t8], dtype=torch.float64, device=dev)
        dist.all_reduce(e8, op=dist.ReduceOp.MAX)
        # ... and no exchange at all: what the per-tick collective costs against it (same process, same buffers)
        def run0(n):
            for _ in range(n):
                solver.solve_batch_device(B, x0.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), bcast, u0g8[0][0].data_ptr(),
                                          status_ptr=status.data_ptr(), stream=stream.cuda_stream)
            dist.barrier()
            torch.cuda.synchronize(dev)
        run0(16)
        t0_ = time.perf_counter()
        run0(n8)
        e0 = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=dev)
        dist.all_reduce(e0, op=dist.ReduceOp.MAX)
        if rank == 0:
            line.setdefault("secondary", {})["no_exchange"] = dict(
                value=world * B / (float(e0.item()) / n8), unit="solves/s", ms_per_step=1e3 * float(e0.item()) / n8,
                what="the same solves with no collective at all (upper bound for any exchange scheme)")
        if rank == 0:
            line.setdefault("secondary", {})["gather_every_8"] = dict(
                value=world * B / (float(e8.item()) / n8), unit="solves/s", ms_per_step=1e3 * float(e8.item()) / n8,
                what="u0 of 8 consecutive ticks share one asynchronous RCCL all-gather (fewer, larger collectives)")
    if rank == 0:
        if world == 1 and not args.no_secondary and args.mapping == "team" and not args.condensed:
            sec = secondary_rows(args, B, N, local, dev, tdt, npdt, torch, _lib, NmpcOcpSolver, x0_h, yref, yref_e, bcast)
            line.setdefault("secondary", {}).update(sec)
            if args.dtype == "f64":
                line["latency_us"] = latency_config1(local, _lib)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
