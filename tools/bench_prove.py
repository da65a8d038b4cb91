#!/usr/bin/env python3
"""Groth16 prove benchmark on the JoinSplit-shaped synthetic circuit (BASELINE.json configs[0]/[3] stand-in; NOT the
Zeth circuit -- see zeth_amd/synth_circuit.py).  Real key from fixed secrets, fixed (r, s), proof compared bit for bit
with the CPU oracle's on the same key and checked with the pairing verifier.  Prints one JSON line."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("ZETH_GPU_HW_QUEUES", "32")  # see bench.py; before HIP initialises
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="alt-bn128")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--constraints", type=int, default=0, help="exact constraint count (overrides --scale)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--in-flight", type=int, default=4, help="proofs kept in flight on the device for the throughput figure")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--pool-dir", default=None, help="with --through-pool: write the C++ driver's input files here and keep them")
    ap.add_argument("--through-pool", action="store_true",
                    help="also prove through the C++ host module: K caller threads on one device_proving_key_pool "
                         "(build/host_adapter_test bench_pool), the proof compared byte for byte with this script's")
    args = ap.parse_args()
    print(json.dumps(measure(args)), flush=True)


def r1cs_file_bytes(cs):
    """The constraint system in Zeth's r1cs_write_bytes format (libzeth/serialization/r1cs_serialization.tcc:199-275):
    u64 primary, u64 auxiliary, u64 constraints, then per constraint a, b, c as u32 term count + (u64 index, Fr big-endian)."""
    import numpy as np

    recs, rps = [], []
    for k in range(3):
        cols = np.asarray(cs.col[k], dtype=np.uint64)
        rec = np.zeros((len(cols), 40), dtype=np.uint8)
        rec[:, :8] = cols.view(np.uint8).reshape(-1, 8)
        cache = {}
        vals = cs.val[k]
        be = bytearray(32 * len(vals))
        for t, v in enumerate(vals):
            b = cache.get(v)
            if b is None:
                b = cache[v] = int(v).to_bytes(32, "big")
            be[32 * t:32 * t + 32] = b
        rec[:, 8:] = np.frombuffer(bytes(be), dtype=np.uint8).reshape(-1, 32)
        recs.append(rec)
        rps.append(np.asarray(cs.row_ptr[k], dtype=np.int64).tolist())
    parts = [np.array([cs.num_inputs, cs.num_variables - cs.num_inputs, cs.num_constraints], dtype=np.uint64).tobytes()]
    for i in range(cs.num_constraints):
        for k in range(3):
            lo, hi = rps[k][i], rps[k][i + 1]
            parts.append((hi - lo).to_bytes(4, "little"))
            parts.append(recs[k][lo:hi].tobytes())
    return b"".join(parts)


def through_pool(args, cs, primary, aux):
    """groth16 proofs through include/libzeth_gpu's device_proving_key_pool from K caller threads (the C++ boundary a
    prover_server would bind): key from the same secrets by the C++ generator, (r, s) as above.  Runs BEFORE this process
    touches the GPU: a second process with a live HIP context on the same device (idle, but holding its hardware queues)
    halves the driver's throughput with four proofs in flight (228 against 440 proofs/s, measured)."""
    import contextlib
    import subprocess
    import tempfile

    driver = os.path.join(ROOT, "build", "host_adapter_test")
    K = max(1, int(args.in_flight or 1))
    keep = getattr(args, "pool_dir", None)  # keep the driver's input files there (to run it under a profiler afterwards)
    with (contextlib.nullcontext(keep) if keep else tempfile.TemporaryDirectory()) as d:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "r1cs"), "wb") as f:
            f.write(r1cs_file_bytes(cs))
        full = list(primary) + list(aux)
        with open(os.path.join(d, "asg"), "wb") as f:
            f.write(len(full).to_bytes(8, "little") + b"".join(int(v).to_bytes(32, "big") for v in full))
        p = subprocess.run([driver, "bench_pool", args.curve, os.path.join(d, "r1cs"), os.path.join(d, "asg"), str(len(primary)),
                            str(K), str(max(4, args.steps)), str(K), os.path.join(d, "proof")], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ))
        if p.returncode != 0:
            raise SystemExit("bench_pool failed (%d): %s %s" % (p.returncode, p.stdout, p.stderr))
        if p.stderr:
            sys.stderr.write(p.stderr)
        res = json.loads(p.stdout.strip().splitlines()[-1])
        got = open(os.path.join(d, "proof"), "rb").read()
    res["_proof_bytes"] = got
    return res


def cpu_prove(cid, cs, extra, primary, aux, rb, sb, threads):
    """The CPU oracle's proof (oracle/zeth_cpu.cpp, oz_prove) on the key material generate_setup_from_secrets returned
    with return_scalars=True: (proof bytes in the C ABI's plain layout, [witness map s, MSM s, total s])."""
    import numpy as np

    import cpu_oracle
    import zeth_amd.lib as zl

    cpu_oracle.lib().oz_set_threads(threads)
    keep = []

    def vp(b):
        cb = ctypes.create_string_buffer(b, max(1, len(b)))
        keep.append(cb)
        return ctypes.cast(cb, ctypes.c_void_p)

    pl = extra["plain"]
    pts = extra["points"]
    views = []
    for k in range(3):
        vals = b"".join(v.to_bytes(32, "little") for v in cs.val[k])
        views.append(cpu_oracle.CsrView(cs.row_ptr[k].ctypes.data, cs.col[k].ctypes.data, vp(vals)))
    views = (cpu_oracle.CsrView * 3)(*views)
    bidx = np.array(extra["b_idx"], dtype=np.uint32)
    enc = lambda g, P: vp(cpu_oracle.enc_point(cid, g, P))  # noqa: E731
    view = cpu_oracle.PkView(enc(1, pts[0]), enc(1, pts[1]), enc(2, pts[2]), enc(1, pts[3]), enc(2, pts[4]),
                             vp(pl["A"]), cs.num_variables + 1, bidx.ctypes.data, vp(pl["B2"]), vp(pl["B1"]), len(bidx),
                             vp(pl["H"]), cs.domain_size() - 1, vp(pl["L"]), cs.num_variables - cs.num_inputs)
    full = vp(zl.encode_scalars(primary) + zl.encode_scalars(aux))
    cout = ctypes.create_string_buffer(8 * zl.fq_bytes(cid))
    sec = (ctypes.c_double * 3)()
    rc = cpu_oracle.lib().oz_prove(cid, ctypes.byref(view), cs.num_inputs, cs.num_variables, cs.num_constraints, views,
                                   full, rb, sb, threads, cout, sec)
    assert rc == 0
    return cout.raw, [sec[0], sec[1], sec[2]]


def measure(args, ctx=None):
    import zeth_amd
    import zeth_amd.lib as zl
    from zeth_amd.keygen import CURVE_INFO, generate_setup_from_secrets
    from zeth_amd.synth_circuit import joinsplit_shaped

    cid = 0 if args.curve == "alt-bn128" else 1
    info = CURVE_INFO[cid]
    t = time.time()
    cs, primary, aux = joinsplit_shaped(info.r, seed=1, scale=args.scale, constraints=getattr(args, "constraints", 0))
    t_circ = time.time() - t
    pool_res = None
    if getattr(args, "through_pool", False):
        if ctx is not None:
            raise SystemExit("--through-pool must run before this process creates its device context")
        pool_res = through_pool(args, cs, primary, aux)
    if ctx is None:
        ctx = zeth_amd.Context(0)
    t = time.time()
    dpk, vk, extra = generate_setup_from_secrets(ctx, cid, cs, t=0x5A4554485F746175, alpha=0x616C706861, beta=0x62657461,
                                                 delta=0x64656C7461, return_scalars=True)
    t_key = time.time() - t
    snark = zeth_amd.Groth16Snark(ctx, cid)
    # full-width randomness, as Fr::random_element() gives libsnark's prover (the host tail's five scalar multiplications
    # cost in proportion to the bit length: 128-bit values, as used up to round 2, halve it)
    r_rand = 0x1234567890ABCDEF1234567890ABCDEF0FEDCBA9876543210FEDCBA987654321 % info.r
    s_rand = 0x0FEDCBA0987654321FEDCBA09876543211234567890ABCDEF1234567890ABCDE % info.r
    n = zl.fq_bytes(cid)
    pb = ctypes.create_string_buffer(zl.encode_scalars(primary), 32 * len(primary))
    ab = ctypes.create_string_buffer(zl.encode_scalars(aux), 32 * len(aux))
    out = ctypes.create_string_buffer(8 * n)
    rb, sb = zl.int_to_le(r_rand, 32), zl.int_to_le(s_rand, 32)

    def prove_once():
        zl._check(ctx._L.zg_prove(dpk._h, pb, len(primary), ab, len(aux), zl.FORM_PLAIN, rb, sb, zl.FORM_PLAIN, out))

    for _ in range(args.warmup):
        prove_once()
    ctx.synchronize()
    times = []
    for _ in range(args.steps):
        t = time.perf_counter()
        prove_once()
        times.append(time.perf_counter() - t)
    raw = out.raw
    proof = (zl.decode_point(cid, 1, raw[: 2 * n]), zl.decode_point(cid, 2, raw[2 * n: 6 * n]), zl.decode_point(cid, 1, raw[6 * n:]))
    if os.environ.get("ZG_BENCH_PROVE_SAMPLES"):
        print("samples_ms", " ".join("%.2f" % (x * 1e3) for x in times), file=sys.stderr)
    times.sort()
    # Throughput with K proofs in flight (BASELINE configs[4], one device's share): K host threads, each with a
    # clone of the resident key (shared bases, own scratch and streams), every proof checked against the serial one.
    inflight = {}
    K = int(getattr(args, "in_flight", 0) or 0)
    if K > 1:
        import threading

        clones = [dpk] + [dpk.clone() for _ in range(K - 1)]
        per_thread = max(4, args.steps)
        outs = [ctypes.create_string_buffer(8 * n) for _ in range(K)]
        errs = []
        barrier = threading.Barrier(K + 1)

        def worker(k):
            try:
                zl._check(ctx._L.zg_prove(clones[k]._h, pb, len(primary), ab, len(aux), zl.FORM_PLAIN, rb, sb, zl.FORM_PLAIN, outs[k]))
                barrier.wait()
                for _ in range(per_thread):
                    zl._check(ctx._L.zg_prove(clones[k]._h, pb, len(primary), ab, len(aux), zl.FORM_PLAIN, rb, sb, zl.FORM_PLAIN, outs[k]))
                    if outs[k].raw != raw:
                        errs.append("thread %d: proof differs from the serial proof" % k)
                        break
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
                try:
                    barrier.abort()
                except Exception:  # noqa: BLE001
                    pass

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
        for th in threads:
            th.start()
        barrier.wait()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        wall = time.perf_counter() - t0
        for c in clones[1:]:
            c.free()
        if errs:
            raise SystemExit("in-flight proving failed: " + "; ".join(errs))
        inflight = {"proofs_in_flight": K, "proofs": K * per_thread, "wall_ms": wall * 1e3,
                    "proofs_per_s_per_gpu": K * per_thread / wall, "all_equal_to_serial_proof": True}
    res = {
        "metric": "groth16_prove_joinsplit_shaped", "curve": args.curve,
        "constraints": cs.num_constraints, "variables": cs.num_variables, "domain": cs.domain_size(),
        "ms_per_proof_median": times[len(times) // 2] * 1e3, "ms_per_proof_min": times[0] * 1e3,
        "proofs_per_s_per_gpu": 1.0 / times[len(times) // 2],
        "timing": "assignment in host memory -> proof in host memory; key resident on device",
        "setup_s": {"circuit": t_circ, "keygen": t_key},
    }
    if inflight:
        res["in_flight"] = inflight
    if not args.no_cpu:
        import cpu_oracle

        threads = cpu_oracle.max_threads()
        cpu_raw, sec = cpu_prove(cid, cs, extra, primary, aux, rb, sb, threads)
        res["bit_exact_vs_cpu_oracle"] = cpu_raw == raw
        res["cpu_baseline"] = {"ms_per_proof": sec[2] * 1e3, "witness_map_ms": sec[0] * 1e3, "msm_ms": sec[1] * 1e3,
                               "cores": threads, "kind": "port",
                               "note": "C++ restatement of the libsnark prover (BDLO12 multi_exp with 0/1 filter, OpenMP), "
                                       "all host cores"}
        res["speedup_vs_cpu"] = sec[2] / times[len(times) // 2]
    if not args.no_verify:
        from zeth_model import BLS12_377, BN254, VerificationKey, verify

        C = BN254 if cid == 0 else BLS12_377
        mvk = VerificationKey()
        mvk.alpha_g1, mvk.beta_g2, mvk.delta_g2, mvk.ABC_g1 = vk.alpha_g1, vk.beta_g2, vk.delta_g2, vk.ABC_g1
        res["verifies"] = bool(verify(C, mvk, primary, proof))
    dpk.free()
    if pool_res is not None:
        from zeth_model import BLS12_377, BN254, proof_to_bytes

        pool_res["proof_equals_ctypes_path"] = pool_res.pop("_proof_bytes") == proof_to_bytes(BN254 if cid == 0 else BLS12_377, proof)
        res["through_cpp_pool"] = pool_res
    return res


if __name__ == "__main__":
    main()
