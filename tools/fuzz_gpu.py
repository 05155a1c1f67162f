"""Randomized differential run of the GPU library against the oracle (development aid, not part of the test suite).
env SECONDS (default 240), SEED.  Prints a line per 50 cases and every mismatch; exit code 1 on any mismatch."""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
import oracle_binding

pkg = entry.load_package(); lib = pkg._abi.lib(); orc = oracle_binding.load()
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
deadline = time.time() + float(os.environ.get("SECONDS", "240"))
GOLD = pkg.NTT_MODULUS
bad = 0; cases = 0
primes = {}


def prime_for(n, bits):
    key = (n, bits)
    if key not in primes:
        primes[key] = orc.L.oracle_largest_prime_1mod(2 * n, bits)
    return primes[key]


def fill(q, shape):
    kind = rng.integers(0, 5)
    if kind == 0: return np.full(shape, q - 1, dtype=np.uint64)
    if kind == 1: return rng.integers(0, 4, size=shape, dtype=np.uint64)
    if kind == 2:
        pool = np.array([0, 1, q - 1, q - 2, q // 2, q // 2 + 1], dtype=np.uint64)
        return pool[rng.integers(0, pool.size, size=shape)]
    return rng.integers(0, q, size=shape, dtype=np.uint64)


def report(what, **kw):
    global bad
    bad += 1
    print("MISMATCH", what, kw, flush=True)


while time.time() < deadline:
    cases += 1
    which = rng.integers(0, 15)
    if which < 5:            # negacyclic transforms, both flavours
        logn = int(rng.integers(1, 18)); n = 1 << logn
        bits = int(rng.integers(max(logn + 3, 14), 61))
        q = prime_for(n, bits)
        if q == 0: continue
        batch = int(rng.integers(1, 40 if logn <= 12 else 6))
        mode = int(rng.integers(0, 2))
        lib.lsr_set_arith_mode(mode)
        ctx = pkg.NttContext(q, n); lib.lsr_set_arith_mode(0)
        a = fill(q, (batch, n))
        f = ctx.forward_batch(a)
        if not np.array_equal(f, orc.ntt_forward(q, n, a)): report("fwd", q=q, n=n, batch=batch, mode=mode)
        if not np.array_equal(ctx.inverse_batch(f), a): report("inv", q=q, n=n, batch=batch, mode=mode)
        g = ctx.inverse_batch(a)
        if not np.array_equal(g, orc.ntt_inverse(q, n, a)): report("inv-of-arbitrary", q=q, n=n, batch=batch, mode=mode)
        b = fill(q, n)
        if not np.array_equal(ctx.mul_pointwise(a[0], b), orc.mul_pointwise(q, n, a[0], b)): report("pointwise", q=q, n=n)
        ctx.close()
    elif which < 7:          # cyclic transforms over the prover's field
        logn = int(rng.integers(1, 18)); n = 1 << logn
        batch = int(rng.integers(1, 20 if logn <= 12 else 4))
        t = pkg.CyclicNtt(n)
        x = fill(GOLD, (batch, n))
        ev = t.forward(x)
        if not np.array_equal(ev, np.stack([orc.cyclic_forward(r, GOLD, t.omega) for r in x])): report("cyclic fwd", n=n, batch=batch)
        if not np.array_equal(t.inverse(ev), x): report("cyclic inv", n=n, batch=batch)
        t.close()
    elif which < 9:          # quotients
        logm = int(rng.integers(0, 10)); m = 1 << logm
        batch = int(rng.integers(1, 30))
        a = fill(GOLD, (batch, m)); b = fill(GOLD, (batch, m))
        c = np.array([[int(x) * int(y) % GOLD for x, y in zip(ra, rb)] for ra, rb in zip(a, b)], dtype=np.uint64)
        for i in range(batch):
            if rng.integers(0, 4) == 0:
                j = int(rng.integers(0, m)); c[i, j] = (int(c[i, j]) + 1) % GOLD
        plan = pkg.QuotientPlan(m)
        quot, lens = plan.quotient_batch(a, b, c)
        for i in range(batch):
            w, ln = orc.quotient(a[i], b[i], c[i])
            if lens[i] != ln or (ln and not np.array_equal(quot[i], w)): report("quotient", m=m, i=i, ln=ln, got=int(lens[i]))
        plan.close()
    elif which == 9:         # fused matrix-vector commitment pipeline at the two-pass degrees (given and device-sampled blinding)
        import torch
        logn = 16 if rng.integers(0, 4) else 17
        n = 1 << logn; k = int(rng.integers(1, 5))
        q = prime_for(n, 44)
        batch = int(rng.integers(1, 12)) if rng.integers(0, 2) else int(rng.integers(60, 70)) if logn == 16 else int(rng.integers(28, 40))
        lctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=int(rng.integers(1, 2**62)))
        a_hat = lctx.public_matrix()
        r = fill(q, (batch, k, n))
        seeds = rng.integers(1, 2**62, size=batch, dtype=np.uint64)
        st = torch.cuda.current_stream().cuda_stream
        d_r = torch.from_numpy(r.view(np.int64)).cuda(); d_u = torch.empty_like(d_r)
        sampled = bool(rng.integers(0, 2))
        picks = sorted({0, batch - 1, int(rng.integers(0, batch))})
        def blinding(j):
            e = np.stack([orc.sample_gaussian_seeded(n, 3.19, int(seeds[j]), 5, i) for i in range(k)])
            return np.where(e < 0, e + q, e).astype(np.uint64)
        if sampled:
            rc = lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), None, d_u.data_ptr(), batch, seeds.ctypes.data, st)
            e1 = {j: blinding(j) for j in picks}
        else:
            e_all = fill(q, (batch, k, n))
            d_e = torch.from_numpy(e_all.view(np.int64)).cuda()
            rc = lctx._lib.lsr_mlwe_matvec_batch_device(lctx.handle, d_r.data_ptr(), d_e.data_ptr(), d_u.data_ptr(), batch, None, st)
            e1 = {j: e_all[j] for j in picks}
        torch.cuda.synchronize()
        if rc != 0: report("fused matvec rc", n=n, k=k, batch=batch, rc=rc)
        for j in picks:
            if not np.array_equal(d_u[j].cpu().numpy().view(np.uint64), orc.mlwe_matvec(q, n, k, a_hat, r[j], e1[j])):
                report("fused matvec", n=n, k=k, batch=batch, j=j, sampled=sampled)
        if not np.array_equal(d_r.cpu().numpy().view(np.uint64), r): report("fused matvec clobbered r", n=n, k=k)
        lctx.close(); del d_r, d_u
    elif which == 10:        # R1CS-level prover: random sparse systems with a constructed satisfying witness
        m = 1 << int(rng.integers(0, 8)); free = int(rng.integers(1, 12)); nv = free + m
        a, b, c = [], [], []
        for i in range(m):
            for mat in (a, b):
                for col in rng.choice(free + i, size=min(int(rng.integers(1, 5)), free + i), replace=False):
                    mat.append((i, int(col), int(rng.integers(0, 2**64, dtype=np.uint64))))
            c.append((i, free + i, 1))
        batch = int(rng.integers(1, 8))
        ws = []
        for _ in range(batch):
            z = [int(x) for x in rng.integers(0, GOLD, size=free, dtype=np.uint64)] + [0] * m
            for i in range(m):
                az = sum((v % GOLD) * z[col] for (r, col, v) in a if r == i) % GOLD
                bz = sum((v % GOLD) * z[col] for (r, col, v) in b if r == i) % GOLD
                z[free + i] = az * bz % GOLD
            if rng.integers(0, 3) == 0: z[free + int(rng.integers(0, m))] ^= 1
            ws.append(z)
        ws = np.array(ws, dtype=np.uint64)
        pr = pkg.R1csProver(m, nv, a, b, c)
        ea, eb, ec = pr.compute_constraint_evals(ws)
        quot, lens = pr.quotient_batch(ws)
        for i in range(batch):
            oa, ob, oc = (orc.sparse_mul_vec(mat, m, ws[i], GOLD) for mat in (a, b, c))
            if not (np.array_equal(ea[i], oa) and np.array_equal(eb[i], ob) and np.array_equal(ec[i], oc)): report("r1cs evals", m=m, i=i)
            w, ln = orc.quotient(oa, ob, oc)
            if lens[i] != ln or (ln and not np.array_equal(quot[i], w)): report("r1cs quotient", m=m, i=i)
        pr.close()
    elif which == 11:        # batch commit, batch verify, linear combination
        n = 1 << int(rng.integers(3, 13)); k = int(rng.integers(1, 4))
        q = [17592169062401, 17592186044417][int(rng.integers(0, 2))]
        seed_key = int(rng.integers(1, 2**62))
        ctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=seed_key)
        cnt = int(rng.integers(1, 6)); ml = int(rng.integers(1, min(n, 64) + 1))
        msgs = rng.integers(0, ctx.plain_modulus, size=(cnt, ml), dtype=np.uint64)   # words >= t never open (commitment.cpp:223-226)
        seeds = rng.integers(1, 2**62, size=cnt, dtype=np.uint64)
        if cnt > 1 and rng.integers(0, 2): seeds[1] = seeds[0]                     # a reused seed: the blinding is message-bound
        coms = pkg.Commitment.batch(ctx, msgs, seeds)
        for i in range(cnt):
            if not np.array_equal(coms[i].as_words(), orc.lwe_commit(q, n, k, 3.19, seed_key, msgs[i], int(seeds[i]))): report("batch commit", n=n, k=k, i=i)
        res = pkg.verify_openings_batch(ctx, coms, msgs)
        if res != [1] * cnt: report("batch verify", n=n, k=k, res=res)
        flat = pkg.Commitment.batch_words(ctx, msgs, seeds)
        if any(not np.array_equal(flat[i], coms[i].as_words()) for i in range(cnt)): report("flat commit", n=n, k=k)
        spoiled = msgs.copy(); victim = int(rng.integers(0, cnt)); spoiled[victim, 0] ^= np.uint64(1)
        out = np.full(cnt, 9, dtype=np.int32)
        if lib.lsr_lwe_verify_opening_batch_flat(ctx.handle, flat.ctypes.data, spoiled.ctypes.data, ml, cnt, out.ctypes.data) != 0: report("flat verify rc")
        want = np.ones(cnt, dtype=np.int32); want[victim] = 0
        if not np.array_equal(out, want): report("flat verify", n=n, k=k, got=out.tolist(), victim=victim)
        coeffs = [int(x) for x in rng.integers(0, 8, size=cnt)]
        if rng.integers(0, 2): coeffs[0] = int(ctx.plain_modulus) - int(rng.integers(1, 40))     # a small negative number (centred representative)
        if rng.integers(0, 3) == 0: coeffs[-1] = int(rng.integers(8, 60))
        comb = pkg.Commitment.linear_combine(ctx, coms, coeffs)
        rc, want = orc.lwe_linear_combine(q, n, k, 3.19, seed_key, [cm.as_words() for cm in coms], coeffs)
        if rc != 0 or not np.array_equal(comb.as_words(), want): report("linear combine", n=n, k=k, cnt=cnt)
        for cm in coms + [comb]: cm.free()
        ctx.close()
    elif which in (12, 13):  # whole commitments / openings, device-resident rows: tile pipeline (n = 4096), fused (2^16, 2^17), general (others)
        import torch
        logn = int([12, 12, 12, 16, 16, 17, 10, 13][int(rng.integers(0, 8))]); n = 1 << logn
        k = int(rng.integers(1, 5))
        q = prime_for(n, 44)
        batch = int(rng.integers(1, 70)) if logn <= 13 else int(rng.integers(1, 6)) if rng.integers(0, 3) else int(rng.integers(30, 40))
        seed_key = int(rng.integers(1, 2**62))
        ctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=seed_key)
        t = ctx.plain_modulus
        ml = int([1, 5, n, n + 2, int(rng.integers(1, n + 1))][int(rng.integers(0, 5))])
        msgs = rng.integers(0, t, size=(batch, ml), dtype=np.uint64)
        seeds = rng.integers(1, 2**62, size=batch, dtype=np.uint64)
        keys = np.zeros((batch, 4), dtype=np.uint64)
        if lib.lsr_lwe_commit_keys(ctx.handle, msgs.ctypes.data, ml, batch, seeds.ctypes.data, keys.ctypes.data) != 0: report("commit keys rc")
        words = lib.lsr_lwe_commitment_words(ctx.handle)
        st = torch.cuda.current_stream().cuda_stream
        d_msgs = torch.from_numpy(msgs.view(np.int64)).cuda(); d_keys = torch.from_numpy(keys.view(np.int64)).cuda()
        d_rows = torch.zeros((batch, words), dtype=torch.int64, device="cuda")
        if lib.lsr_lwe_commit_rows_device(ctx.handle, d_msgs.data_ptr(), ml, batch, d_keys.data_ptr(), d_rows.data_ptr(), st) != 0: report("commit rows rc")
        torch.cuda.synchronize()
        picks = sorted({0, batch - 1, int(rng.integers(0, batch))})
        host = d_rows[picks].cpu().numpy().view(np.uint64)
        for i, j in enumerate(picks):
            if not np.array_equal(host[i], orc.lwe_commit(q, n, k, 3.19, seed_key, msgs[j], int(seeds[j]))):
                report("commit rows", n=n, k=k, batch=batch, ml=ml, j=j, pipeline=lib.lsr_lwe_pipeline(ctx.handle).decode())
        vl = min(ml, n)
        claims = np.ascontiguousarray(msgs[:, :vl]).copy(); victim = int(rng.integers(0, batch)); claims[victim, int(rng.integers(0, vl))] ^= np.uint64(1)
        d_claims = torch.from_numpy(claims.view(np.int64)).cuda()
        res = torch.full((batch,), 9, dtype=torch.int32, device="cuda")
        if lib.lsr_lwe_verify_rows_device(ctx.handle, d_rows.data_ptr(), d_claims.data_ptr(), vl, batch, res.data_ptr(), st) != 0: report("verify rows rc")
        torch.cuda.synchronize()
        want = np.ones(batch, dtype=np.int32); want[victim] = 0
        if not np.array_equal(res.cpu().numpy(), want): report("verify rows", n=n, k=k, batch=batch, vl=vl, victim=victim)
        ctx.close(); del d_rows, d_msgs, d_keys
    elif which == 14:        # per-commitment keys derived on the device = the host derivation; long flat batches (device keys behind the host call)
        import torch
        n = 1 << int(rng.integers(3, 13)); k = int(rng.integers(1, 4))
        seed_key = int(rng.integers(1, 2**62))
        ctx = pkg.LweContext(pkg.Params(q=17592186044417, n=n, k=k, sigma=3.19), key_seed=seed_key)
        ml = int(rng.integers(0, n + 40)); batch = int(rng.integers(1, 300))
        msgs = rng.integers(0, 2**64, size=(batch, ml), dtype=np.uint64) if rng.integers(0, 2) else rng.integers(0, ctx.plain_modulus, size=(batch, ml), dtype=np.uint64)
        seeds = rng.integers(1, 2**64, size=batch, dtype=np.uint64)
        want = ctx.commit_keys(msgs, seeds)
        d_msgs = torch.from_numpy(msgs.view(np.int64)).cuda() if ml else None
        d_keys = torch.zeros((batch, 4), dtype=torch.int64, device="cuda")
        ctx.commit_keys_device(d_msgs.data_ptr() if ml else None, ml, seeds, d_keys.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if not np.array_equal(d_keys.cpu().numpy().view(np.uint64), want): report("device keys", n=n, k=k, ml=ml, batch=batch)
        if ml and batch * min(ml, n) >= 2**16:
            small = msgs % np.uint64(ctx.plain_modulus)
            rows = pkg.Commitment.batch_words(ctx, small, seeds)
            j = int(rng.integers(0, batch))
            if not np.array_equal(rows[j], orc.lwe_commit(17592186044417, n, k, 3.19, seed_key, small[j], int(seeds[j]))): report("long flat batch", n=n, k=k, ml=ml, batch=batch, j=j)
        ctx.close()
    else:                    # commitments
        n = 1 << int(rng.integers(1, 13)); k = int(rng.integers(1, 5))
        q = [12289, 17592186044417, 17592169062401, prime_for(n, int(rng.integers(41, 61)))][int(rng.integers(0, 4))]
        if q == 0: continue
        seed_key = int(rng.integers(1, 2**62))
        ctx = pkg.LweContext(pkg.Params(q=q, n=n, k=k, sigma=3.19), key_seed=seed_key)
        msg = rng.integers(0, 2**21, size=int(rng.integers(0, n + 3)), dtype=np.uint64)   # some words above t on purpose
        seed = int(rng.integers(1, 2**62))
        com = pkg.Commitment(ctx, msg, seed)
        want = orc.lwe_commit(q, n, k, 3.19, seed_key, msg % np.uint64(q), seed)     # Commitment::new reduces mod the REQUESTED modulus (commitment.rs:33-36)
        if not np.array_equal(com.as_words(), want): report("commit", q=q, n=n, k=k, len=len(msg))
        shown = (msg % np.uint64(q))[:n]
        t = np.uint64(ctx.plain_modulus)
        if pkg.verify_opening_with_context(ctx, com, shown) != bool((shown < t).all()): report("verify raw words", q=q, n=n, k=k)
        if not pkg.verify_opening_with_context(ctx, com, shown % t): report("verify", q=q, n=n, k=k)
        com.free(); ctx.close()
    if cases % 50 == 0:
        print(f"{cases} cases, {bad} mismatches, {deadline - time.time():.0f} s left", flush=True)
print(f"done: {cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
