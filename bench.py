#!/usr/bin/env python3
"""Headline benchmark: BN254 (alt_bn128) G1 multi-scalar multiplication of 2^24 points (BASELINE.json
configs[1]/[2]) through the C ABI, inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one MSM over the whole 2^24-point job.  With N ranks the job is sharded by point range (rank g
holds bases [g n/N, (g+1) n/N) resident), each rank reduces its shard to one affine point, the N partial points
(64 B each) are all-gathered over RCCL as raw bytes and added on the host: strong scaling, the reference's
"sum over OpenMP chunks" (SURVEY.md 8e).  Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The prover keeps five HIP streams busy per proof and several proofs in flight; with HIP's default of 4 hardware
# queues per process they serialise (measured: 8.0 -> 6.6 ms per proof and 150 -> 250 proofs/s with 4 in flight, from 4
# to 32 queues).  Must be set before the HIP runtime initialises; no effect on the single-stream MSM headline.
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("ZETH_GPU_HW_QUEUES", "32")

MAC_PER_POINT_BN254_G1 = 21760  # 16 windows x 10 field mults x 136 32-bit MACs
PEAK_MAC_PER_S = 3.85e13  # v_mad_u64_u32 issue rate (4.08 cycles per wave64 instruction) measured on MI355X by
# tools/peak_valu.hip, profiles/r1_peak_valu.jsonl; the microarchitecture guide gives no integer-multiply figure
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md


def cpu_baseline(bases, scalars_dev, sample_n):
    """Times the CPU restatement of libff multi_exp<BDLO12> (oracle/zeth_cpu.cpp) on the first sample_n
    points of the same workload, OpenMP chunks = threads, and checks it against the GPU on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes

    import cpu_oracle
    import zeth_amd.lib as zl

    # every thread OpenMP gives this process: omp_get_max_threads(), what libsnark's MULTICORE build uses for its
    # chunks.  On the GPU box OMP_NUM_THREADS is the job's CPU share (16 of a machine whose nproc is many times that).
    threads = cpu_oracle.max_threads()
    cpu_oracle.lib().oz_set_threads(threads)
    ab = 64
    buf = ctypes.create_string_buffer(sample_n * ab)
    zl._check(zl.load_library().zg_bases_read(bases._h, 0, sample_n, zl.FORM_PLAIN, buf))
    sc = scalars_dev.download(32 * sample_n)
    cpu_pt, sec = cpu_oracle.msm_bytes(0, 1, buf.raw, sc, sample_n, method=0, chunks=threads)
    gpu_pt = bases.msm(scalars_dev, n=sample_n)
    if cpu_pt != gpu_pt:
        raise SystemExit("cpu_baseline: GPU and CPU oracle disagree on the %d-point sample" % sample_n)
    return {
        "value": sample_n / sec,
        "unit": "points/s",
        "cores": threads,
        "kind": "port",
        "sample": "first 2^%d points of the same bases/scalars, libff-style multi_exp<BDLO12> with %d OpenMP chunks, "
        "%.2f s; result equal to the GPU's on that sample" % (sample_n.bit_length() - 1, threads, sec),
        "host": host_description(),
        "build": "oracle/Makefile: -O2 -march=x86-64-v3 -madx -fopenmp (the reference: -O2 -march=native, CMakeLists.txt:197-199; "
        "native is not usable for a library built in one container and run on another host); portable unsigned __int128 "
        "limbs, not libff's x86 assembly",
    }


def host_description():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
            "sched_affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}


def ntt_report(ctx, zeth_amd):
    """Evaluation-domain transform (BN254 Fr) at the prover's sizes: kernel time from HIP events on the library's stream,
    as fractions of the HBM roofline (64 m algorithmic bytes per transform) and of the multiply peak."""
    import ctypes

    pmc, pmc_src = {}, None
    for name in ("r3_pmc_ntt.json", "r2_pmc_ntt.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            pmc_src = "profiles/" + name
            break
        except (OSError, ValueError):
            pass
    out = []
    for lg in (19, 22):
        m = 1 << lg
        buf = zeth_amd.DeviceBuffer.random_scalars(ctx, zeth_amd.ALT_BN128, m, 99)

        def run(inv, coset):
            zeth_amd.lib._check(ctx._L.zg_ntt(ctx._h, 0, ctypes.c_void_p(buf.ptr), lg, inv, coset, zeth_amd.lib.FORM_MONT, 1))
            return sum(ms for name, ms in ctx.last_timings() if name.startswith("ntt_"))

        run(1, 0)
        per = {}
        for inv, coset, name in ((1, 0, "iFFT"), (0, 1, "cosetFFT"), (1, 1, "icosetFFT")):
            ks = sorted(run(inv, coset) for _ in range(7))
            per[name] = ks[len(ks) // 2]
        k_ms = sum(per.values()) / len(per)
        buf.free()
        out.append({"log_m": lg, "kernels_ms": k_ms, "per_variant_ms": per,
                    "hbm": {"algorithmic_bytes": 64 * m, "GBps": 64 * m / (k_ms * 1e-3) / 1e9, "frac": 64 * m / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    "mac": {"per_transform": (m // 2) * lg * 136, "frac": (m // 2) * lg * 136 / (k_ms * 1e-3) / PEAK_MAC_PER_S},
                    "traffic": dict(pmc[str(lg)], source=pmc_src + " (collected in separate rocprofv3 --pmc runs, not in this run)") if str(lg) in pmc else None})
    return {"curve": "alt-bn128 Fr", "transforms": out,
            "note": "the prover's three variants, mean of their medians; 64 m bytes = one read and one write of the vector "
            "(SURVEY 8d), (m/2) log2 m butterflies x 136 MAC; traffic = FETCH_SIZE/WRITE_SIZE per transform "
            "(profiles/r2_pmc_ntt.json, separate rocprofv3 --pmc passes) when collected"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--cpu-log-n", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prove", action="store_true", help="skip the JoinSplit-shaped Groth16 prove measurement")
    ap.add_argument("--no-table", action="store_true", help="windowed MSM on the bare bases (no fixed-base table)")
    ap.add_argument("--no-windowed", action="store_true", help="skip the bare-bases (windowed) leg")
    ap.add_argument("--no-g2", action="store_true", help="skip the G2 MSM leg")
    ap.add_argument("--g2-log-n", type=int, default=22)
    ap.add_argument("--no-ntt", action="store_true", help="skip the evaluation-domain transform report")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # torch is plumbing for the multi-rank exchange only (RCCL all-gather of 64-byte partial results).  A single
    # rank never imports it: every buffer is owned by the library, and the process then runs on the system ROCm
    # runtime alone (DESIGN.md "torch and rocprofv3").
    torch = dist = None
    backend = os.environ.get("ZETH_BENCH_BACKEND", "nccl")
    xdev = "cuda" if backend == "nccl" else "cpu"  # where the exchanged bytes live
    if world > 1:
        import torch
        import torch.distributed as dist

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the library has no CPU path")
        # ZETH_BENCH_BACKEND=gloo is a rehearsal knob for a box with fewer GPUs than ranks (ranks then share devices and
        # exchange over the CPU); the driver's runs use RCCL, one rank per GPU
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank %= torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)

    import zeth_amd
    from zeth_amd.synth import closed_form_scalar, random_scalars_host

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from zeth_model import BN254  # checker only: expected point from the closed form

    C = BN254
    n_total = 1 << args.log_n
    if n_total % world:
        raise SystemExit("2^%d points do not split over %d ranks" % (args.log_n, world))
    n_local = n_total // world
    first = rank * n_local
    t0, d = 0x5A4554480000000100000000DEADBEEF, 0x5A45544800000002FEEDFACECAFEF00D

    ctx = zeth_amd.Context(local_rank)
    ctx.set_profiling(True)
    bases = zeth_amd.Bases.generate(ctx, zeth_amd.ALT_BN128, zeth_amd.G1, n_local, (t0 + first * d) % C.r, d)
    # The bases of a proving key are fixed: their multiples 2^(j c) P_i are built once (outside the timed region, like
    # loading the key) and stay resident; every MSM then needs 12-13 mixed additions per scalar instead of 16.
    table = None
    if not args.no_table:
        t_build = time.perf_counter()
        bases.precompute()
        ctx.synchronize()
        cw, levels = bases.table_info()
        table = {"window_bits": cw, "levels": levels, "bytes_per_rank": n_local * levels * 64,
                 "build_s": time.perf_counter() - t_build}
    digits = table["levels"] if table else 16
    seed = 0x5A455448 + (rank << 32)
    scalars = zeth_amd.DeviceBuffer.random_scalars(ctx, zeth_amd.ALT_BN128, n_local, seed)  # resident in HBM

    from zeth_amd.distributed import all_gather_sum_bytes
    from zeth_amd.lib import decode_point

    def step():
        part = bases.msm_bytes(scalars, n_local)  # raw affine bytes: no integer decoding inside the timed loop
        return all_gather_sum_bytes(0, 1, part, xdev) if world > 1 else part

    def fence():
        ctx.synchronize()  # the library's stream (every zg_msm call has already returned its result to the host)
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    result = None
    for _ in range(args.warmup):
        result = step()
    fence()
    accum_ms = []
    phase_ms = {}
    t_start = time.perf_counter()
    for _ in range(args.steps):
        result = step()
        for name, ms in ctx.last_timings():
            phase_ms[name] = phase_ms.get(name, 0.0) + ms
            if name == "accum":
                accum_ms.append(ms)
    fence()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of the whole job (outside the timed region): closed form (sum a_i (t0 + i d)) G
    words = random_scalars_host(n_local, C.r, seed)
    import numpy as np

    probe = np.frombuffer(scalars.download(32 * 4096), dtype=np.uint64).reshape(-1, 4)
    if not np.array_equal(probe, words[:4096]):
        raise SystemExit("bench.py: host mirror of the scalar generator disagrees with the device")
    k_local = closed_form_scalar(words, (t0 + first * d) % C.r, d, C.r)
    if world > 1:
        ks = [None] * world
        dist.all_gather_object(ks, k_local)
        k = sum(ks) % C.r
    else:
        k = k_local
    expect = C.G1.mul(k, C.G1.gen)
    result = decode_point(0, 1, result)
    if result != expect:
        again = decode_point(0, 1, step())
        raise SystemExit("bench.py: MSM result does not match the closed form (rerun equal to first: %s, rerun correct: %s)"
                         % (again == result, again == expect))

    cpu_base = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(bases, scalars, min(n_local, 1 << args.cpu_log_n))

    def timed_leg(step_fn, steps):
        """`steps` calls of step_fn between fences, max over ranks: (ms per step, k_accum ms per step, last result)"""
        res = step_fn()
        fence()
        acc = []
        t_a = time.perf_counter()
        for _ in range(steps):
            res = step_fn()
            acc += [ms for name, ms in ctx.last_timings() if name == "accum"]
        fence()
        el = time.perf_counter() - t_a
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=xdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el / steps * 1e3, sum(acc) / max(1, len(acc)), res

    # The same job on the BARE bases (no table): what libff::multi_exp computes per call -- 16 signed 16-bit windows.
    windowed = None
    if table is not None and not args.no_windowed:
        bare = zeth_amd.Bases.generate(ctx, zeth_amd.ALT_BN128, zeth_amd.G1, n_local, (t0 + first * d) % C.r, d)

        def step_w():
            part = bare.msm_bytes(scalars, n_local)
            return all_gather_sum_bytes(0, 1, part, xdev) if world > 1 else part

        w_ms, w_acc, w_res = timed_leg(step_w, max(3, args.steps // 2))
        if decode_point(0, 1, w_res) != expect:
            raise SystemExit("bench.py: windowed MSM result does not match the closed form")
        bare.free()
        w_achieved = n_local * 16 * (MAC_PER_POINT_BN254_G1 // 16) / (w_acc * 1e-3)
        windowed = {"ms_per_step": w_ms, "points_per_s": n_total / (w_ms * 1e-3), "digits_per_scalar": 16,
                    "roofline": {"kernel": "k_accum", "kernel_ms": w_acc, "achieved": w_achieved / 1e12, "peak": PEAK_MAC_PER_S / 1e12,
                                 "unit": "TMAC/s", "frac": w_achieved / PEAK_MAC_PER_S},
                    "verified": "closed form", "note": "bare bases, no precomputed table: the libff::multi_exp equivalent"}

    # BASELINE configs[2]'s second leg: G2 MSM of 2^22 points, sharded by point range like the G1 job
    g2 = None
    if not args.no_g2:
        n2_total = 1 << args.g2_log_n
        n2 = n2_total // world
        f2 = rank * n2
        seed2 = 0x47320000 + (rank << 32)
        bases2 = zeth_amd.Bases.generate(ctx, zeth_amd.ALT_BN128, zeth_amd.G2, n2, (t0 + f2 * d) % C.r, d)
        if not args.no_table:
            bases2.precompute()
        sc2 = zeth_amd.DeviceBuffer.random_scalars(ctx, zeth_amd.ALT_BN128, n2, seed2)

        def step_g2():
            part = bases2.msm_bytes(sc2, n2)
            return all_gather_sum_bytes(0, 2, part, xdev) if world > 1 else part

        g_ms, g_acc, g_res = timed_leg(step_g2, max(3, args.steps // 2))
        k2 = closed_form_scalar(random_scalars_host(n2, C.r, seed2), (t0 + f2 * d) % C.r, d, C.r)
        if world > 1:
            ks2 = [None] * world
            dist.all_gather_object(ks2, k2)
            k2 = sum(ks2) % C.r
        if decode_point(0, 2, g_res) != C.G2.mul(k2, C.G2.gen):
            raise SystemExit("bench.py: G2 MSM result does not match the closed form")
        cw2, lv2 = bases2.table_info()
        g_digits = lv2 if lv2 else 16
        g_achieved = n2 * g_digits * (65280 // 16) / (g_acc * 1e-3)
        g2 = {"metric": "bn254_g2_msm_2^%d" % args.g2_log_n, "ms_per_step": g_ms, "points_per_s": n2_total / (g_ms * 1e-3),
              "fixed_base_table": {"window_bits": cw2, "levels": lv2} if lv2 else None, "verified": "closed form",
              "roofline": {"kernel": "k_accum (G2)", "kernel_ms": g_acc, "achieved": g_achieved / 1e12, "peak": PEAK_MAC_PER_S / 1e12,
                           "unit": "TMAC/s", "frac": g_achieved / PEAK_MAC_PER_S,
                           "note": "4080 MAC per G2 mixed addition (SURVEY 8d: Fq2 product = 3 Fq products)"}}
        sc2.free()
        bases2.free()

    ntt = ntt_report(ctx, zeth_amd) if rank == 0 and not args.no_ntt else None

    # second headline number (BASELINE.json metric): Groth16 proofs/s on the JoinSplit-shaped circuit, every rank a
    # replica with its own resident key (no collective); outside the timed MSM region above
    prove = None
    if not args.no_prove:
        import types

        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_prove

        bases.free()
        pargs = types.SimpleNamespace(curve="alt-bn128", scale=1.0, steps=40, warmup=2, constraints=0, in_flight=4,
                                      no_cpu=(rank != 0 or args.no_cpu_baseline), no_verify=(rank != 0))
        prove = bench_prove.measure(pargs, ctx)
        # node throughput = proofs kept in flight on every device (BASELINE configs[4]); latency is reported beside it
        rate = prove.get("in_flight", {}).get("proofs_per_s_per_gpu", prove["proofs_per_s_per_gpu"])
        if world > 1:
            tr = torch.tensor([rate], dtype=torch.float64, device=xdev)
            dist.all_reduce(tr, op=dist.ReduceOp.SUM)
            rate = float(tr.item())
        prove["proofs_per_s_node"] = rate
        prove["n_gpus"] = world

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = sum(accum_ms) / len(accum_ms)
        # work the kernel actually does: one mixed addition (10 field mults x 136 MACs = 21760 / 16) per (point, digit)
        achieved = n_local * digits * (MAC_PER_POINT_BN254_G1 // 16) / (k_ms * 1e-3)
        # algorithmic HBM bytes of the accumulation kernel: per (point, digit) one 64 B base + one 4 B sorted entry
        hbm_bytes = n_local * digits * 68
        # HBM bytes per k_accum launch from the PMC passes of this same workload (the counters cannot be read from
        # inside this process); only quoted for the size they were collected at
        traffic = traffic_source = None
        for name in ("r3_pmc_hbm_traffic.json", "r2_pmc_hbm_traffic.json", "r1_pmc_hbm_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]["k_accum"]
                if n_local == 1 << 24 and pmc.get("digits_per_scalar", 16) == digits:
                    traffic = pmc["FETCH_SIZE_bytes_per_launch"] + pmc["WRITE_SIZE_bytes_per_launch"]
                    traffic_source = "profiles/" + name + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this workload; NOT measured in this run)"
                break
            except (OSError, KeyError, ValueError):
                continue
        out = {
            "metric": "bn254_g1_msm_2^%d_throughput" % args.log_n,
            "value": n_total / (elapsed / args.steps),
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "BN254 G1 MSM, 2^%d points P_i=(t0+i*d)G resident in HBM, scalars uniform mod r resident in HBM, "
                "result affine on host" % args.log_n,
                "sharding": "point range over %d rank(s); 64 B partials all-gathered" % world,
                "verified": "closed form (sum a_i t_i) G",
                "fixed_base_table": table,
            },
            "roofline": {
                "bound": "valu_int32_mul",
                "kernel": "k_accum (bucket accumulation, XYZZ mixed add)",
                "achieved": achieved / 1e12,
                "peak": PEAK_MAC_PER_S / 1e12,
                "unit": "TMAC/s",
                "frac": achieved / PEAK_MAC_PER_S,
                "kernel_ms": k_ms,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "traffic_unit": "bytes per k_accum launch (FETCH_SIZE + WRITE_SIZE)",
                "algorithmic_bytes": hbm_bytes,
                "hbm_algorithmic_GBps": hbm_bytes / (k_ms * 1e-3) / 1e9,
                "hbm_frac": hbm_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "digits_per_scalar": digits,
                "note": "MSM is integer-multiply bound (SURVEY 8d): achieved = 1360 MAC per mixed addition x points x digits per "
                "scalar / kernel time (16 digits = SURVEY's 21760 MAC/point; the fixed-base table lowers the digits, i.e. the "
                "work, not this fraction); peak = measured v_mad_u64_u32 rate",
            },
            "phases_ms": {k2: v / args.steps for k2, v in phase_ms.items()},
        }
        out["hip_runtime"] = zeth_amd.runtime_info()  # which libamdhip64 / libhsa-runtime64 served this run
        if windowed is not None:
            out["windowed"] = windowed
            out["windowed_ms_per_step"] = windowed["ms_per_step"]
        if g2 is not None:
            out["g2_msm"] = g2
        if ntt is not None:
            out["ntt"] = ntt
        if prove is not None:
            out["joinsplit_prove"] = prove
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
