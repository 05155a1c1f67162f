#!/usr/bin/env python3
"""Headline benchmark: BASELINE config 2 — batched forward+inverse negacyclic NTT, n = 2^16, 4096 polynomials per
GPU, device-resident, q16 = 17592182243329 (the 44-bit prime 17592169062401 of the reference cannot host
n = 2^16: SURVEY.md F5) — plus config 3 (rank-4 Module-LWE matrix–vector commitment, 1024 witness vectors).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one forward pass + one inverse pass over the rank's 4096 polynomials (8192 transforms).  Ranks are
independent (weak scaling, no data-path collective — SURVEY.md §8(e)); the only communication is the barrier and
the MAX-reduction of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

Q16 = 17592182243329
N = 65536
NTT_BYTES = 2 * N * 8                 # algorithmic bytes of one transform: read n*8 + write n*8
HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--polys", type=int, default=4096, help="polynomials per GPU (config 2: 4096)")
    ap.add_argument("--commits", type=int, default=1024, help="witness vectors per GPU (config 3: 1024)")
    ap.add_argument("--rank", type=int, default=4, help="module rank k of the commitment workload")
    ap.add_argument("--no-commit", action="store_true", help="skip the config-3 section")
    ap.add_argument("--no-quotient", action="store_true", help="skip the quotient-polynomial section (SURVEY.md §8(f) rank 2)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-polys", type=int, default=8192, help="CPU baseline sample: polynomials transformed fwd+inv on one core (~10 s)")
    ap.add_argument("--check-polys", type=int, default=64, help="forward outputs compared word for word with the CPU oracle (outside the timed region)")
    return ap.parse_args()


def do_verify_long(fctx, lib, d_rows, d_msgs, msg_len, count, d_res, stream):
    """every row opens to its message (lsr_lwe_verify_rows_device); outside any timed region"""
    import torch
    assert lib.lsr_lwe_verify_rows_device(fctx.handle, d_rows.data_ptr(), d_msgs.data_ptr(), msg_len, count, d_res.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    return bool((d_res == 1).all().item())


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample_polys, gpu_forward_sample, seed_base, commit_check):
    """The oracle (a restatement of the SEAL Harvey NTT the reference calls) timed on ONE host core — the
    reference's execution model is single-threaded.  kind = "port": the reference itself cannot be
    built here (SEAL absent).  The same leg is the CHECKER of the timed GPU run (outside the timed region): polynomial i
    is splitmix64(seed_base + i) mod q exactly as on the device, and the oracle's forward outputs of the first
    len(gpu_forward_sample) polynomials are compared word for word with what the GPU produced."""
    import __graft_entry__ as entry
    orc = entry.load_oracle()
    h = orc.ntt_handle(Q16, N)
    chunk = 64
    done = 0
    spent = 0.0
    matches = True
    while done < sample_polys:
        buf = np.concatenate([orc.splitmix(seed_base + done + i, Q16, N) for i in range(chunk)])
        orig = buf.copy() if done == 0 else None
        t0 = time.perf_counter()
        orc.L.oracle_ntt_forward_batch(h, buf.ctypes.data, chunk)
        spent += time.perf_counter() - t0
        if done < len(gpu_forward_sample):
            take = min(chunk, len(gpu_forward_sample) - done)
            matches = matches and np.array_equal(buf.reshape(chunk, N)[:take], gpu_forward_sample[done:done + take])
        t0 = time.perf_counter()
        orc.L.oracle_ntt_inverse_batch(h, buf.ctypes.data, chunk)
        spent += time.perf_counter() - t0
        if orig is not None:
            matches = matches and np.array_equal(buf, orig)
        done += chunk
    single = {"value": 2 * done / spent, "unit": "NTT/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
              "sample": f"{done} polys x (fwd+inv), n=2^16, q={Q16}, inputs splitmix64(0xDEADBEEF+i), single thread, {spent:.1f} s",
              "gpu_forward_matches_cpu": bool(matches), "gpu_forward_polys_compared": int(len(gpu_forward_sample))}
    # the same port, one polynomial stream per thread on the box's CPU share for one GPU (16 threads): informational
    import threading
    threads, per_thread = 16, 4
    bufs = [orc.splitmix(0xC0FFEE + i, Q16, chunk * N) for i in range(threads)]

    def work(b):
        for _ in range(per_thread):
            orc.L.oracle_ntt_forward_batch(h, b.ctypes.data, chunk)
            orc.L.oracle_ntt_inverse_batch(h, b.ctypes.data, chunk)

    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(b,)) for b in bufs]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt_mt = time.perf_counter() - t0
    single["multithread"] = {"value": 2 * threads * per_thread * chunk / dt_mt, "unit": "NTT/s", "cores": threads}
    if commit_check is not None:      # config 3 checker: sampled witness vectors of the timed commit batch against the oracle
        k, a_hat, picks, r_rows, e1_rows, u_rows = commit_check
        ok = True
        t0 = time.perf_counter()
        for j in range(len(picks)):
            ok = ok and np.array_equal(orc.mlwe_matvec(Q16, N, k, a_hat, r_rows[j], e1_rows[j]), u_rows[j])
        dt_c = time.perf_counter() - t0
        single["commit"] = {"value": len(picks) / dt_c, "unit": "commits/s", "cores": 1, "gpu_commit_matches_cpu": bool(ok),
                            "vectors_compared": [int(x) for x in picks]}
    return single


def latest_profile_json(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    with open(files[-1]) as f:
        return json.load(f)


def provenance_module():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import provenance
    finally:
        sys.path.pop(0)
    return provenance


def pmc_profile(pattern):
    """The newest committed PMC artefact profiles/<pattern> — but only if it was measured on THIS library's sources
    (tools/provenance.py stamp): PMC counters need rocprofv3, so the run cannot collect them itself; a figure from another build
    is reported as null, never silently."""
    prof = latest_profile_json(pattern)
    if prof is None:
        return None, "no profile"
    if not provenance_module().matches(prof):
        return None, "profile was measured on other sources than the running library"
    return prof, "profile matches the running library's sources"


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU; the library has no CPU fallback")
    # BENCH_REHEARSE_SHARED_GPU=1 (development): every rank of a torch.distributed.run launch uses device 0 and the job's barriers
    # and reductions run over gloo — the N > 1 control flow of this file (matched collectives, the config-4 leg with waiting
    # ranks, rank 0's line) on a one-GPU box.  The line then says so and its value is not a scaling figure.
    rehearsal = os.environ.get("BENCH_REHEARSE_SHARED_GPU") == "1"
    if rehearsal:
        local = 0
        os.environ["LAMBDA_SNARK_DEVICE"] = "0"
    torch.cuda.set_device(local)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run (any world size)
    reduce_device = "cpu" if rehearsal else "cuda"
    cpu_group = None
    if use_dist:
        # gloo announces its connections on the process's stdout ("[Gloo] Rank 0 is connected to ..."): the job's stdout carries
        # rank 0's JSON line and nothing else, so file descriptor 1 points at stderr while the groups are set up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            if world > 1:    # a CPU-side group: ranks that only WAIT (config-4 leg) must not spin in a collective kernel on their GPU
                cpu_group = dist.new_group(backend="gloo")
                dist.barrier(group=cpu_group)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    pkg = entry.load_package()
    ctx = pkg.NttContext(Q16, N, device=local)
    stream = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0xDEADBEEF + rank)
    lib = pkg._abi.lib()
    # polynomial i = splitmix64(0xDEADBEEF + i) mod q, one draw per coefficient (global index i)
    seed_base = 0xDEADBEEF + rank * args.polys
    polys = torch.empty((args.polys, N), dtype=torch.int64, device="cuda")
    assert lib.lsr_fill_splitmix_device(polys.data_ptr(), args.polys, N, seed_base, Q16, stream) == 0
    torch.cuda.synchronize()
    reference_copy = polys[:8].clone()
    # checker sample (outside the timed region): forward outputs of the first polynomials, compared with the CPU oracle below
    n_check = min(args.check_polys, args.polys)
    ctx.forward_device(polys.data_ptr(), n_check, stream)
    torch.cuda.synchronize()
    gpu_forward_sample = polys[:n_check].cpu().numpy().view(np.uint64).copy()
    ctx.inverse_device(polys.data_ptr(), n_check, stream)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ctx.forward_device(polys.data_ptr(), args.polys, stream)
        ctx.inverse_device(polys.data_ptr(), args.polys, stream)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    verified = bool(torch.equal(polys[:8], reference_copy))        # fwd∘inv identity survived every step

    # per-direction device time with HIP events on the launch stream (roofline figure)
    def event_time(fn, reps):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e-3

    reps = max(5, min(args.steps, 20))
    t_fwd = event_time(lambda: ctx.forward_device(polys.data_ptr(), args.polys, stream), reps)
    t_inv = event_time(lambda: ctx.inverse_device(polys.data_ptr(), args.polys, stream), reps)
    fwd_rate, inv_rate = args.polys / t_fwd, args.polys / t_inv

    extra = {"fwd_ntt_per_s": fwd_rate, "inv_ntt_per_s": inv_rate, "fwd_ms_per_batch": t_fwd * 1e3, "inv_ms_per_batch": t_inv * 1e3,
             "verified_roundtrip": verified,
             "arith": ("u64 residues mod q; products by exact FP64-FMA Barrett (q < 2^45, DESIGN.md §4), canonical u64 in and out"
                       if ctx.uses_f64 else "u64 Harvey/Shoup lazy butterflies")}
    # the reference's own algorithm (SEAL Harvey butterflies, Shoup products on u64) on the GPU, same array, for comparison
    lib.lsr_set_arith_mode(1)
    ctx_u64 = pkg.NttContext(Q16, N, device=local)
    lib.lsr_set_arith_mode(0)
    ctx_u64.forward_device(polys.data_ptr(), args.polys, stream)
    t_fwd_u64 = event_time(lambda: ctx_u64.forward_device(polys.data_ptr(), args.polys, stream), max(3, reps // 2))
    extra["u64_flavour_fwd_ntt_per_s"] = args.polys / t_fwd_u64
    extra["u64_flavour_roofline_frac"] = args.polys / t_fwd_u64 * NTT_BYTES / (HBM_PEAK_GBS * 1e9)
    ctx_u64.close()

    # ---- config 3: rank-k Module-LWE matrix–vector commitment u = INTT(A^T NTT(r)) + e1 ----
    commit_check = None
    commit_roofline = None
    del polys
    if not args.no_commit:
        torch.cuda.empty_cache()
        k = args.rank
        lctx = pkg.LweContext(pkg.Params(q=Q16, n=N, k=k, sigma=3.19), key_seed=0xC0DE + 1, device=local)
        # r_j uniform from splitmix64(0xC0FFEE + j), e1 from the seeded CDT sampler (sigma = 3.19)
        first = rank * args.commits
        r = torch.empty((args.commits, k, N), dtype=torch.int64, device="cuda")
        assert lib.lsr_fill_splitmix_device(r.data_ptr(), args.commits, k * N, 0xC0FFEE + first, Q16, stream) == 0
        e1 = torch.empty_like(r)
        e1_seeds = (np.arange(first + 1, first + args.commits + 1, dtype=np.uint64) * np.uint64(0x9E3779B9))
        assert lib.lsr_lwe_sample_blinding_device(lctx.handle, e1.data_ptr(), args.commits, e1_seeds.ctypes.data, stream) == 0
        u = torch.empty_like(r)
        r_work = torch.empty_like(r)

        def commit_step():
            rc = lib.lsr_mlwe_matvec_batch_device(lctx.handle, r_work.data_ptr(), e1.data_ptr(), u.data_ptr(), args.commits, None, stream)
            assert rc == 0

        # every step commits to the SAME witness vectors.  Contexts on the fused pipeline only read r; an unfused context
        # overwrites it with NTT(r), and then the working copy is restored outside the HIP-event bracket of each step.
        r_work.copy_(r)
        commit_step()
        torch.cuda.synchronize()
        r_preserved = bool(torch.equal(r_work[:2], r[:2]) and torch.equal(r_work[-1], r[-1]))
        extra["commit_input_preserved"] = r_preserved

        def timed_commit_steps(count):
            evs = []
            for _ in range(count):
                if not r_preserved:
                    r_work.copy_(r)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); commit_step(); b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            return [a.elapsed_time(b) * 1e-3 for a, b in evs]

        timed_commit_steps(max(1, args.warmup // 2))
        barrier()
        csteps = max(3, args.steps // 2)
        c_times = timed_commit_steps(csteps)
        c_el = float(np.sum(c_times))
        if use_dist:
            t = torch.tensor([c_el], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c_el = float(t.item())
        commit_bytes = 3 * k * N * 8      # read r + read e1 + write u (6 291 456 B at k = 4)
        commits_per_s = world * args.commits * csteps / c_el
        per_gpu = commits_per_s / world
        pmc, pmc_note = pmc_profile("r*_pmc_commit_traffic.json")
        pmc = pmc or {}
        commit_roofline = {"bound": "hbm", "achieved": per_gpu * commit_bytes / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": per_gpu * commit_bytes / (HBM_PEAK_GBS * 1e9),
                           "traffic": (pmc.get("bytes_per_commit") or 0) * args.commits or None,
                           "traffic_source": pmc_note,
                           "algorithmic_bytes": commit_bytes * args.commits, "launch_ms": float(np.median(c_times)) * 1e3,
                           "kernel": pmc.get("kernel", "lsr_mlwe_matvec_batch_device launch sequence (DESIGN.md §5)")}
        extra.update({"commits_per_s": commits_per_s, "commit_rank": k, "commits_per_gpu": args.commits,
                      "commit_roofline_frac": commit_roofline["frac"], "commit_roofline": commit_roofline})
        # the same workload with e1 sampled on the device (no e1 array in HBM: 4 194 304 B per commit algorithmic)
        r_work.copy_(r)
        assert lib.lsr_mlwe_matvec_batch_device(lctx.handle, r_work.data_ptr(), None, u.data_ptr(), args.commits, e1_seeds.ctypes.data, stream) == 0
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            r_work.copy_(r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert lib.lsr_mlwe_matvec_batch_device(lctx.handle, r_work.data_ptr(), None, u.data_ptr(), args.commits, e1_seeds.ctypes.data, stream) == 0
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        extra["commits_per_s_e1_on_device"] = args.commits / float(np.median(ts))
        extra["commit_roofline_frac_e1_on_device"] = extra["commits_per_s_e1_on_device"] * 2 * k * N * 8 / (HBM_PEAK_GBS * 1e9)
        if rank == 0 and not args.no_cpu and world == 1:
            picks = [0, 127, 128, args.commits - 1]
            picks = sorted({p for p in picks if 0 <= p < args.commits})
            tonp = lambda t: t.cpu().numpy().view(np.uint64)
            commit_check = (k, lctx.public_matrix(), picks, [tonp(r[p]) for p in picks], [tonp(e1[p]) for p in picks], [tonp(u[p]) for p in picks])
        lctx.close()
        del r, e1, u, r_work

    # ---- "next" row: NTT-path quotient polynomials, 4096 instances of m = 4096 constraints ----
    if not args.no_quotient:
        torch.cuda.empty_cache()
        qm, qb = 4096, 4096
        plan = pkg.QuotientPlan(qm, device=local)
        field = pkg.CyclicNtt(qm, device=local)
        ea = torch.randint(-2**63, 2**63 - 1, (qb, qm), dtype=torch.int64, device="cuda", generator=gen)
        eb = torch.randint(-2**63, 2**63 - 1, (qb, qm), dtype=torch.int64, device="cuda", generator=gen)
        ec = torch.empty_like(ea)
        assert lib.lsr_ntt_mul_pointwise_device(field.handle, ec.data_ptr(), ea.data_ptr(), eb.data_ptr(), qb * qm, stream) == 0   # satisfied: c = a*b
        quot = torch.empty_like(ea)
        qlen = torch.empty(qb, dtype=torch.int32, device="cuda")

        def quotient_step():
            plan.quotient_device(ea.data_ptr(), eb.data_ptr(), ec.data_ptr(), qb, quot.data_ptr(), qlen.data_ptr(), stream)

        quotient_step()
        t_q = event_time(quotient_step, reps)
        extra.update({"quotients_per_s": qb / t_q, "quotient_constraints_per_s": qb * qm / t_q, "quotient_m": qm,
                      "quotient_all_valid": bool((qlen > 0).all().item()),
                      # algorithmic bytes: read a, b, c evaluations, write the quotient (4 m words per instance)
                      "quotient_roofline_frac": qb * qm * 32 / t_q / (HBM_PEAK_GBS * 1e9)})
        plan.close(); field.close()

    # ---- whole commitments and openings, device-resident (keys, messages and wire rows in HBM; HIP events on the launch stream):
    #      lwe_commit / lwe_verify_opening for a batch (cpp-core/src/commitment.cpp:138-164,200-232) at the reference's parameters
    #      (n = 4096, k = 2: one launch, one workgroup per commitment) and at config 3's shape (n = 2^16, k = 4: sampled inside the
    #      strided rounds).  Algorithmic bytes per commitment = the wire row ((k + 1) n + 5 words), written once / read once. ----
    if not args.no_commit and rank == 0:
        torch.cuda.empty_cache()
        full = {}
        for label, fq, fn_, fk, fb in (("n4096_k2", 17592169062401, 4096, 2, 16384), ("n65536_k4", Q16, N, 4, 1024)):
            fctx = pkg.LweContext(pkg.Params(q=fq, n=fn_, k=fk, sigma=3.19), key_seed=0xC0DE + 3, device=local)
            words = lib.lsr_lwe_commitment_words(fctx.handle)
            msg_len = 16
            rng = np.random.default_rng(3)
            fmsgs = rng.integers(0, fctx.plain_modulus, size=(fb, msg_len), dtype=np.uint64)
            fseeds = rng.integers(1, 2**63, size=fb, dtype=np.uint64)
            fkeys = np.zeros((fb, 4), dtype=np.uint64)
            t0 = time.perf_counter()
            assert lib.lsr_lwe_commit_keys(fctx.handle, fmsgs.ctypes.data, msg_len, fb, fseeds.ctypes.data, fkeys.ctypes.data) == 0
            t_keys = time.perf_counter() - t0
            d_msgs = torch.from_numpy(fmsgs.view(np.int64)).cuda()
            d_keys = torch.from_numpy(fkeys.view(np.int64)).cuda()
            d_rows = torch.empty((fb, words), dtype=torch.int64, device="cuda")
            d_res = torch.zeros(fb, dtype=torch.int32, device="cuda")
            do_commit = lambda: lib.lsr_lwe_commit_rows_device(fctx.handle, d_msgs.data_ptr(), msg_len, fb, d_keys.data_ptr(), d_rows.data_ptr(), stream)
            do_verify = lambda: lib.lsr_lwe_verify_rows_device(fctx.handle, d_rows.data_ptr(), d_msgs.data_ptr(), msg_len, fb, d_res.data_ptr(), stream)
            assert do_commit() == 0 and do_verify() == 0
            t_c = event_time(do_commit, max(3, reps // 2))
            t_v = event_time(do_verify, max(3, reps // 2))
            torch.cuda.synchronize()
            row_bytes = words * 8
            full[label] = {"pipeline": lib.lsr_lwe_pipeline(fctx.handle).decode(), "batch": fb, "row_bytes": row_bytes,
                           "commits_per_s": fb / t_c, "commit_ms_per_batch": t_c * 1e3, "commit_roofline_frac": fb * row_bytes / t_c / (HBM_PEAK_GBS * 1e9),
                           "openings_per_s": fb / t_v, "verify_ms_per_batch": t_v * 1e3, "verify_roofline_frac": fb * row_bytes / t_v / (HBM_PEAK_GBS * 1e9),
                           "all_rows_open": bool((d_res == 1).all().item()),
                           "host_key_derivation_commits_per_s": fb / t_keys}
            # the oracle's words for two of the timed rows (outside the timed region)
            if not args.no_cpu and world == 1:
                orc = entry.load_oracle()
                host_rows = d_rows[[0, fb - 1]].cpu().numpy().view(np.uint64)
                ok = all(np.array_equal(host_rows[i], orc.lwe_commit(fq, fn_, fk, 3.19, 0xC0DE + 3, [int(x) for x in fmsgs[j]], int(fseeds[j])))
                         for i, j in enumerate((0, fb - 1)))
                full[label]["rows_match_cpu_oracle"] = bool(ok)
            # FULL-LENGTH messages (msg_len = n: a polynomial's worth of field elements per commitment, what the Rust prover commits to),
            # keys derived on the device from the device-resident messages (lsr_lwe_commit_keys_device): key schedule + commitments,
            # nothing but the seeds from the host; beside it the host key derivation the device one replaces
            if fn_ == 4096:
                del d_msgs
                long_msgs = torch.randint(0, int(fctx.plain_modulus), (fb, fn_), dtype=torch.int64, device="cuda", generator=gen)
                d_keys2 = torch.empty_like(d_keys)
                do_keys = lambda: lib.lsr_lwe_commit_keys_device(fctx.handle, long_msgs.data_ptr(), fn_, fb, fseeds.ctypes.data, d_keys2.data_ptr(), stream)
                do_long = lambda: lib.lsr_lwe_commit_rows_device(fctx.handle, long_msgs.data_ptr(), fn_, fb, d_keys2.data_ptr(), d_rows.data_ptr(), stream)
                do_both = lambda: (do_keys(), do_long())
                assert do_keys() == 0 and do_long() == 0 and do_verify_long(fctx, lib, d_rows, long_msgs, fn_, fb, d_res, stream)
                t_k = event_time(do_keys, max(3, reps // 2))
                t_kc = event_time(do_both, max(3, reps // 2))
                sample = long_msgs[:512].cpu().numpy().view(np.uint64)
                t0 = time.perf_counter()
                assert lib.lsr_lwe_commit_keys(fctx.handle, sample.ctypes.data, fn_, 512, fseeds.ctypes.data, fkeys.ctypes.data) == 0
                t_host = time.perf_counter() - t0
                full[label]["full_length_messages"] = {
                    "msg_len": fn_, "commits_per_s_keys_and_rows_on_device": fb / t_kc, "device_key_derivation_per_s": fb / t_k,
                    "host_key_derivation_per_s": 512 / t_host, "all_rows_open": True,
                    "device_keys_equal_host_keys": bool(np.array_equal(d_keys2[:512].cpu().numpy().view(np.uint64), fkeys[:512]))}
                del long_msgs, d_keys2
            else:
                del d_msgs
            fctx.close()
            del d_rows, d_keys, d_res
        extra["full_commit"] = full

    # ---- complete commitments at the reference's parameters (n = 4096, k = 2), PCIe included: host-visible throughput of the
    #      additive flat entry points (never the headline value) ----
    if not args.no_commit and rank == 0:
        torch.cuda.empty_cache()
        rctx = pkg.LweContext(pkg.Params(q=17592186044417, n=4096, k=2, sigma=3.19), key_seed=0xC0DE + 2, device=local)
        nb = 2048
        rmsgs = (np.arange(nb * 8, dtype=np.uint64).reshape(nb, 8) * 7919) % 1000003
        rseeds = np.arange(1, nb + 1, dtype=np.uint64)
        rows = np.zeros((nb, pkg._abi.lib().lsr_lwe_commitment_words(rctx.handle)), dtype=np.uint64)
        verdicts = np.zeros(nb, dtype=np.int32)
        flat_lib = pkg._abi.lib()

        def wall(fn, reps=5):
            fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts))

        t_c = wall(lambda: flat_lib.lsr_lwe_commit_batch_flat(rctx.handle, rmsgs.ctypes.data, 8, nb, rseeds.ctypes.data, rows.ctypes.data))
        t_v = wall(lambda: flat_lib.lsr_lwe_verify_opening_batch_flat(rctx.handle, rows.ctypes.data, rmsgs.ctypes.data, 8, nb, verdicts.ctypes.data))
        extra.update({"ref_params_commits_per_s_pcie": nb / t_c, "ref_params_openings_per_s_pcie": nb / t_v,
                      "ref_params_all_verified": bool((verdicts == 1).all())})
        # the same two calls with the rows in page-locked host memory (lsr_host_alloc_pinned): the rows then travel in 32 MiB pieces
        # under the next piece's kernels.  In a fresh process this leg runs at 497-530 K commitments/s (tools/pin_probe.py); inside
        # this process, after PyTorch's caching allocator has returned gigabytes to the driver, the same call takes 5.7 ms instead of
        # 3.9-4.1 (tools/pin_probe2.py reproduces it with nothing but a 2 GiB tensor freed before) — reported as measured
        try:
            pin = pkg.PinnedArray(rows.shape)
            t_cp = wall(lambda: flat_lib.lsr_lwe_commit_batch_flat(rctx.handle, rmsgs.ctypes.data, 8, nb, rseeds.ctypes.data, pin.ptr))
            verdicts[:] = 0
            t_vp = wall(lambda: flat_lib.lsr_lwe_verify_opening_batch_flat(rctx.handle, pin.ptr, rmsgs.ctypes.data, 8, nb, verdicts.ctypes.data))
            extra.update({"ref_params_commits_per_s_pcie_pinned": nb / t_cp, "ref_params_openings_per_s_pcie_pinned": nb / t_vp,
                          "ref_params_pinned_rows_equal": bool(np.array_equal(pin.array, rows)) and bool((verdicts == 1).all())})
            pin.close()
        except Exception as exc:      # informational
            extra["ref_params_pinned_error"] = f"{type(exc).__name__}: {exc}"
        rctx.close()

    # ---- config 4: the config-3 batch cut into contiguous slices over the node's GPUs INSIDE the library (one host thread and
    #      stream per device, device-resident inputs, slices gathered device -> host into one pinned array); run by rank 0 over
    #      every device of the job while the other ranks wait — no data-path collective ----
    def cpu_barrier():
        torch.cuda.synchronize()
        if cpu_group is not None:
            dist.barrier(group=cpu_group)

    def config4_leg():
        k = args.rank
        n_dev = max(1, min(world, lib.lsr_device_count()))
        total = args.commits
        main_ctx = pkg.LweContext(pkg.Params(q=Q16, n=N, k=k, sigma=3.19), key_seed=0xC0DE + 1, device=0)
        twins = [main_ctx] + [main_ctx.replicate(d) for d in range(1, n_dev)]
        parts_r, parts_e = [], []
        for g in range(n_dev):
            first, count = pkg.shard_bounds(total, n_dev, g)
            with torch.cuda.device(g):
                st = torch.cuda.current_stream().cuda_stream
                pr = torch.empty((max(count, 1), k, N), dtype=torch.int64, device=f"cuda:{g}")
                pe = torch.empty_like(pr)
                assert lib.lsr_fill_splitmix_device(pr.data_ptr(), count, k * N, 0xC0FFEE + first, Q16, st) == 0
                sd = (np.arange(first + 1, first + count + 1, dtype=np.uint64) * np.uint64(0x9E3779B9))
                assert lib.lsr_lwe_sample_blinding_device(twins[g].handle, pe.data_ptr(), count, sd.ctypes.data, st) == 0
                torch.cuda.synchronize()
            parts_r.append(pr); parts_e.append(pe)
        pinned = pkg.PinnedArray((total, k, N))
        ptr_r, ptr_e = [p.data_ptr() for p in parts_r], [p.data_ptr() for p in parts_e]
        pkg.sharded_matvec(twins, ptr_r, ptr_e, total, pinned.array)              # warm-up (workspaces, pinned pages)
        walls, stats = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            st = pkg.sharded_matvec_stats(twins, ptr_r, ptr_e, total, pinned.array)
            walls.append(time.perf_counter() - t0); stats.append(st)
        w = float(np.median(walls))
        mid = stats[int(np.argsort(walls)[len(walls) // 2])]          # the per-shard figures of the median run
        result = {"devices": n_dev, "commits": total, "commits_per_s_with_host_gather": total / w, "wall_ms": w * 1e3,
                  "kernel_ms_slowest_shard": max(kk for kk, _ in mid) * 1e3, "wall_ms_slowest_shard": max(ww for _, ww in mid) * 1e3,
                  "per_device": [{"device": g, "vectors": pkg.shard_bounds(total, n_dev, g)[1], "kernel_ms": kk * 1e3, "wall_ms_until_gathered": ww * 1e3,
                                  "gather_GBps": pkg.shard_bounds(total, n_dev, g)[1] * k * N * 8 / ww / 1e9 if ww > 0 else None}
                                 for g, (kk, ww) in enumerate(mid)],
                  "note": "lsr_mlwe_matvec_batch_sharded_stats: device-resident r and e1 per shard; every shard copies one piece of its slice "
                          "to the pinned host array while it computes the next, so its wall time is about max(kernels, gather) + one piece"}
        pinned.close()
        for tctx in twins:
            tctx.close()
        return result

    if not args.no_commit:
        torch.cuda.empty_cache()
        cpu_barrier()                 # the other ranks wait on the CPU (gloo), their GPUs idle, while rank 0 drives every device
        if rank == 0:
            try:
                extra["config4"] = config4_leg()
            except Exception as exc:      # an informational leg must never cost the headline line
                extra["config4"] = {"error": f"{type(exc).__name__}: {exc}"}
        cpu_barrier()

    if rank == 0:
        ntt_pmc, ntt_pmc_note = pmc_profile("r*_roofline_inputs.json")
        extra["provenance"] = provenance_module().provenance()
        transforms = 2 * args.polys * args.steps * world
        value = transforms / elapsed
        achieved = fwd_rate * NTT_BYTES / 1e9
        line = {
            "metric": "degree-2^16 NTTs/sec (forward+inverse, batched, device-resident)",
            "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64",                                 # residues mod q in uint64; the mechanism of the products is in extra.arith
            "data": "synthetic" + (" (REHEARSAL: all ranks on device 0, gloo)" if rehearsal else ""),
            "config": {"workload": "config2: batched forward+inverse negacyclic NTT, n=2^16, 4096 polys/GPU, q=17592182243329 (44-bit)",
                       "polys_per_gpu": args.polys, "ring_degree": N, "modulus": Q16, "parallelism": f"independent batches x{world}, no collectives"},
            # achieved = ALGORITHMIC bytes (1 MiB per transform) / measured time of one forward batch launch sequence
            # (8 chunks x {ntt_strided_round<4>, ntt_tile_forward<12>}); traffic = PMC bytes of the same sequence.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": ((ntt_pmc or {}).get("forward_bytes_per_transform") or 0) * args.polys or None,
                         "traffic_source": ntt_pmc_note,
                         "algorithmic_bytes": NTT_BYTES * args.polys, "launch_ms": t_fwd * 1e3,
                         "kernel": "forward NTT batch = ntt_strided_round<ArithF64,4> + ntt_tile_forward<ArithF64,12> per 512-poly chunk",
                         # what two passes over the array can reach on this part: every residue crosses the L2 <-> memory fabric
                         # twice (32 B where the algorithm counts 16); the measured pure-movement floor of that pattern is
                         # 1.27-1.31 ms per 4096 polynomials (profiles/r02_ubench_move2.txt, r01_ubench_pack.txt)
                         "ceiling_frac": 4096 * NTT_BYTES / 1.272e-3 / (HBM_PEAK_GBS * 1e9),
                         "ceiling_source": "profiles/r02_ubench_move2.txt (two-pass data movement alone: 1.272 ms per 4096 polynomials)"},
            "extra": extra,
        }
        if not args.no_cpu and world == 1:      # reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args.cpu_polys, gpu_forward_sample, seed_base, commit_check)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
