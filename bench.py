#!/usr/bin/env python
"""bench.py -- Groth16 proofs/sec on the synthetic 2^20-constraint chain QAP (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W                      (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one groth16::prove of the 2^20-gate chain circuit (m = 2n+2 wires,
l = 2, roots w^j) per GPU, with CRS, QAP and witness already resident in HBM.  At N > 1 (--mode exchange, the
default) a step is one ROUND of N proofs: rank j runs the SpMV / NTT stage of proof j, RCCL all-to-alls hand every
rank the scalars that multiply its own point range of the four inner products, the rank accumulates them for all N
proofs in grouped MSMs, a second all-to-all returns the 768-byte partial sums to the owners (scaling "weak": per-GPU
work per step does not depend on N).  --mode shard is the latency form (one proof at a time, every rank repeats
the NTT stage, one all-gather; scaling "strong"); --mode replicas runs one independent prover per GPU.

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (dominant kernel, HIP-event
timed inside the library over the timed region) and, at N = 1, `cpu_baseline` (the CPU oracle's
faithful restatement of the reference's prove() timed on this host, single thread like the
reference, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_instance(zk, ctx, log_n, seed, witness="uniform", roots="unity"):
    from zksnark_rs_amd.circuits import chain_rows, chain_weights
    rng = zk.SplitMix64(seed)
    n = 1 << log_n
    m, l, u, v, w = chain_rows(log_n)
    x = rng.fr()
    if witness == "boolean":    # inputs a_k in {0, 1}: half of all wires are bits (one very heavy MSM bucket)
        avals = [rng.next() & 1 for _ in range(n)]
    elif witness == "small":    # 32-bit inputs
        avals = [rng.next() & 0xFFFFFFFF for _ in range(n)]
    else:
        avals = [rng.fr() for _ in range(n)]
    weights = chain_weights(log_n, x, avals)
    td = zk.ints_to_limbs([rng.fr() for _ in range(5)])
    r, s = rng.fr(), rng.fr()
    # the same rows over the roots w^j (the metric's workload) or over ASTParser's roots 1..n
    if roots == "arbitrary":    # affine images a k + b of the integers, handed over as caller data: treated as arbitrary field elements
        import numpy as np
        a, b = rng.fr() | 1, rng.fr()
        k = np.arange(1, n + 1, dtype=object)
        qap = ctx.qap_sparse_roots(zk.ints_to_limbs([int(v_) for v_ in (a * k + b) % zk.R_MODULUS]).reshape(n, 4), m, l, u, v, w)
    else:
        qap = ctx.qap_sparse(log_n, m, l, u, v, w) if roots == "unity" else ctx.qap_sparse_integers(n, m, l, u, v, w)
    crs = ctx.setup(qap, td)      # groth16::setup on the GPU, outside the timed region
    return dict(n=n, m=m, l=l, rows=(u, v, w), qap=qap, crs=crs, weights=weights, td=td, r=r, s=s, log_n=log_n)


def cpu_baseline(zk, ctx, seed, main_inst, full=False):
    """CPU legs of SURVEY 8d, timed on this host in this run (the reference itself is Rust + crate bn and cannot be built here):
    B2, the headline: the oracle's SAME-ALGORITHM path (NTT + Pippenger, bit-identical output) measured DIRECTLY on the metric's
        workload -- one 2^20 proof on all host threads -- and at 2^16 on 1 thread and on all threads (2^20 on one thread takes
        minutes: only with --cpu-baseline full);
    B1: the FAITHFUL restatement of the reference's prove() (dense QAP, schoolbook multiply, long division with an inversion per
        term, one double-and-add per inner-product term; single thread like the reference) measured at 2^6, 2^8, 2^10, 2^12.  It
        cannot run at 2^20 (O(m n) + O(n^2) field work on a dense QAP of 3 m n 32 B = 105 TB), so beyond 2^12 only SURVEY 8d's
        operation count priced with unit costs measured here is given, labelled as an extrapolation."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load()
    t_start = time.time()
    threads = max(1, os.cpu_count() or 1)

    def descs(inst):
        desc = ctx.sparse_desc(inst["log_n"], inst["m"], inst["l"], *inst["rows"])
        return desc, ctx.crs_desc(inst["n"], inst["m"], inst["l"], ctx.crs_download(inst["crs"]))

    def check(inst, proof):
        assert proof == ctx.prove(inst["crs"], inst["qap"], inst["weights"], inst["r"], inst["s"]), "CPU and GPU proofs differ"

    faithful = []
    for log_n in (6, 8, 10, 12):
        inst = build_instance(zk, ctx, log_n, seed + log_n)
        desc, cdesc = descs(inst)
        sec, proof = orc.time_prove_sparse(desc, cdesc, inst["weights"], inst["r"], inst["s"], True, 1)
        check(inst, proof)
        faithful.append((inst["n"], sec))
    k_f, k_i, k_g1, k_g2 = orc.unit_costs()

    def model(n):
        m, l = 2 * n + 2, 2
        return k_f * ((3 * n + 1) * n + n * n + (n - 1) * (n + 1)) + k_i * (n - 1) + k_g1 * (3 * n + m - l + 3) + k_g2 * (n + 1)
    ratios = [t / model(n) for n, t in faithful]
    calib = ratios[-1]
    same = {}
    inst16 = build_instance(zk, ctx, 16, seed + 16)
    desc, cdesc = descs(inst16)
    for th in (1, threads):
        sec, proof = orc.time_prove_sparse_mt(desc, cdesc, inst16["weights"], inst16["r"], inst16["s"], th, 1)
        check(inst16, proof)
        same["2^16, %d thread%s" % (th, "" if th == 1 else "s")] = round(sec, 3)
    del inst16
    desc, cdesc = descs(main_inst)
    legs20 = (threads, 1) if full else (threads,)
    sec20 = None
    for th in legs20:
        sec, proof = orc.time_prove_sparse_mt(desc, cdesc, main_inst["weights"], main_inst["r"], main_inst["s"], th, 1)
        check(main_inst, proof)
        same["2^%d, %d thread%s" % (main_inst["log_n"], th, "" if th == 1 else "s")] = round(sec, 3)
        if th == threads:
            sec20 = sec
    n_main = float(main_inst["n"])
    return {
        "value": 1.0 / sec20, "unit": "proofs/s", "cores": threads, "kind": "port",
        "sample": "ONE whole proof of the metric's workload (2^%d-gate chain circuit) by the oracle's same-algorithm CPU path (NTT + Pippenger, "
                  "%d host threads, bytes equal to the GPU's), measured directly: %.2f s" % (main_inst["log_n"], threads, sec20),
        "same_algorithm_seconds_per_proof": same,
        "reference_algorithm": {
            "what": "oracle's FAITHFUL restatement of the reference's prove() (mod.rs:213-296), 1 thread like the reference, QAP prebuilt",
            "measured_seconds_per_proof": {"2^%d" % (n.bit_length() - 1): round(t, 3) for n, t in faithful},
            "unit_costs": {"fr_mul_add_ns": round(k_f * 1e9, 1), "fr_inverse_us": round(k_i * 1e6, 2), "g1_scalar_mul_us": round(k_g1 * 1e6, 1),
                           "g2_scalar_mul_us": round(k_g2 * 1e6, 1)},
            "measured_over_model": [round(x, 2) for x in ratios],
            "EXTRAPOLATED_seconds_per_proof": {"2^16": round(calib * model(65536.0), 1), "2^20": float("%.3g" % (calib * model(n_main)))},
            "note": "extrapolation = SURVEY 8d's operation count T(n) = k_F (rho n + n^2 + (n-1)(n+1)) + k_I (n-1) + k_G1 (3n+m-l+3) + k_G2 (n+1), rho = 3n+1, "
                    "calibrated at 2^12; the faithful path itself cannot run there (dense QAP = 3 m n 32 B)",
        },
        "wall_s": round(time.time() - t_start, 1),
    }


PMC_KERNEL = {"msm_accumulate_g1": "zk::k_msm_accumulate<zk::Fp<zk::FqParams> >", "msm_accumulate_g2": "zk::k_msm_accumulate<zk::Fq2>"}
PMC_FILES = {20: "r3_pmc_traffic.json", 16: "r3_pmc_traffic_2p16.json"}   # passes exist for the metric's size and for config 3 (2^16)
PMC_ACC_FILES = {20: "r3_pmc_acc.json", 16: "r3_pmc_acc_2p16.json"}

# VALU issue ceiling of gfx950 for the two instruction classes of the multiplier, from tools/ubench_valu.hip (>= 5 ms kernels, in-kernel
# shader / wall clocks, cross-checked with SQ_INSTS_VALU and GRBM_GUI_ACTIVE: profiles/r2_ubench_valu.txt, r2_ubench_valu_pmc.txt):
#   64-bit / integer-multiply class (v_mad_u64_u32, v_mad_i64_i32, v_mul_lo_u32, v_lshl_add_u64, v_ashrrev_i64, carry adds): 4 cycles
#   per wave-instruction and SIMD -> 1024 SIMDs x 2.4 GHz / 4 = 614 G/s (measured 585 at 8 waves/SIMD, 546 at 3, clock 2.36-2.40 GHz);
#   plain 32-bit class (v_and, v_add_u32, v_mov, shifts): 2 cycles -> 1229 G/s (measured 1062).
VALU_PEAK_G = {"slow": 1024 * 2.4 / 4.0, "fast": 1024 * 2.4 / 2.0}
VALU_MEASURED_G = {"slow": 585.0, "fast": 1062.0}


def pmc_acc(name, log_n=20):
    """(wave-instructions per addition, share of the 4-cycle class, source) of an accumulation kernel, MEASURED: tools/profile_round.sh ->
    tools/pmc_acc_summary.py -> profiles/r3_pmc_acc.json (SQ_INSTS_VALU per addition; the class share = SQ_INSTS_VALU_INT64 /
    SQ_INSTS_VALU of the executed instructions).  None when the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_ACC_FILES[log_n])) as f:
            k = json.load(f)["kernels"][name]
        return float(k["wave_instr_per_addition"]), float(k["share_4_cycle_class"]), k
    except Exception:
        return None


def msm_window(count, opt=0, g2=False):
    """The window size the library uses for a product of `count` points -- a mirror of msm_auto_window (csrc/msm_impl.hpp) and of the
    msm_window_bits option's encoding (csrc/crs.hip: c | 100 big + small | + 10000 g2), so that the bench line can count additions."""
    if g2 and opt // 10000 > 0:
        return opt // 10000
    o = opt % 10000
    if 0 < o < 100:
        return o
    if o >= 100:
        return o // 100 if count >= (1 << 21) - 8 else o % 100
    if count + 8 >= (1 << 21) or (g2 and count + 8 >= (1 << 20)):
        return 20
    lg = max(count, 1).bit_length() - 1
    return 17 if lg >= 17 else 16 if lg >= 16 else 15 if lg >= 14 else 13 if lg >= 11 else 8


def pmc_traffic(name, log_n=20):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this round's build (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this same command, tools/pmc_summary.py -> profiles/r2_pmc_traffic.json; FETCH_SIZE corrected
    as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be collected inside this process: null when the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILES[log_n])) as f:
            return json.load(f)["kernels"][PMC_KERNEL[name]]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--mode", choices=["exchange", "shard", "replicas"], default="exchange",
                    help="N > 1: exchange = every proof's inner products sharded over the ranks by point ranges, the SpMV/NTT stage done "
                         "once per proof by its owner rank, scalars and partial sums moved by all-to-all (a step is one round of N "
                         "proofs); shard = the latency form (one proof at a time, every rank repeats the NTT stage, one all-gather); "
                         "replicas = independent provers")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic, 1 GPU only: time rank 0's share of a W-way window-sharded proof (no collective) instead of whole proofs")
    ap.add_argument("--transport", choices=["zk", "zk-gloo", "torch"], default="zk",
                    help="N > 1: zk = the communicator and the exchange pipeline inside libzkgpu.so (RCCL linked into the library: zk_comm_*, "
                         "zk_mgpu_*; torch.distributed is never initialised, the RCCL id travels through a TCP store); torch = the round-1 driver "
                         "in Python over torch.distributed collectives; zk-gloo (also what --backend gloo selects) = the same C pipeline over a "
                         "caller-supplied gloo transport, for functional runs of several ranks on ONE GPU")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="--transport torch: torch.distributed backend; gloo + several ranks on ONE GPU is a functional check of the N > 1 path only")
    ap.add_argument("--shard", choices=["windows", "points"], default="points",
                    help="what a rank owns of each inner product in --mode shard: Pippenger windows w = rank (mod N), or the "
                         "point range [count rank / N, count (rank+1) / N) with every window (5 %% faster at N = 8: 15 windows "
                         "do not divide by 8, and a rank sorts only its own scalars)")
    ap.add_argument("--serialize", action="store_true",
                    help="measurement mode: no kernel overlap (stand-alone kernel durations); needs the ZK_MEASURE build of the library "
                         "(make -C zksnark_rs_amd/csrc measure; ZKGPU_LIB=zksnark_rs_amd/libzkgpu_measure.so)")
    ap.add_argument("--witness", choices=["uniform", "boolean", "small"], default="uniform",
                    help="distribution of the chain circuit's inputs a_k (the metric is quoted on 'uniform')")
    ap.add_argument("--seed", type=int, default=20260929)
    ap.add_argument("--depth", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="proofs in flight per GPU (zk_prove_submit/zk_prove_wait); 1 = synchronous zk_prove_dev; "
                         "0 = 2 for whole proofs, 4 for the per-rank shares of a sharded proof")
    ap.add_argument("--witness-from", choices=["hbm", "pinned", "pageable"], default="hbm",
                    help="N = 1: where each proof's witness is when the proof is submitted.  hbm (the metric: inputs resident); pinned / "
                         "pageable = host memory handed to zk_prove_submit_host (the PCIe-inclusive rate quoted in DESIGN.md)")
    ap.add_argument("--batch", type=int, default=1,
                    help="N = 1: proofs per zk_prove_batch_submit (grouped inner products; for circuits of 2^16 gates and fewer, where "
                         "a lone proof is bound by launch latency).  The metric's 2^20 workload is quoted with --batch 1")
    ap.add_argument("--roots", choices=["unity", "integers", "arbitrary"], default="unity",
                    help="QAP domain: unity = w^j (the metric's workload); integers = 1..n, what ASTParser gives the same circuit "
                         "(zk_qap_upload_sparse_integers); arbitrary = caller-supplied field elements (zk_qap_upload_sparse_roots: the prover "
                         "interpolates per proof by a sub-product tree).  N = 1, no batches; secondary measurements, no CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-times", action="store_true",
                    help="event-time every launch group of a proof, not only the bucket accumulations (kernel_ms_per_proof then lists them "
                         "all; costs ~1 %% of the rate)")
    ap.add_argument("--no-profile", action="store_true",
                    help="no per-kernel event timing inside the library during the timed region (two event records per launch): the line then "
                         "carries no roofline object; `host` reports the time the host spent enqueueing")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="zk_set_option(KEY, VALUE) before the run: measurement switches, refused unless the loaded library is the ZK_MEASURE build")
    ap.add_argument("--latency", action="store_true",
                    help="one proof at a time (--depth 1), no per-kernel event timing (two extra API calls per launch, which small "
                         "circuits feel), no CPU baseline: ms_per_step is the latency of a lone zk_prove_dev call")
    ap.add_argument("--cpu-baseline", choices=["default", "full"], default="default",
                    help="full: also the same-algorithm CPU path at 2^20 on ONE thread (minutes)")
    args = ap.parse_args()
    if args.latency:
        args.depth, args.no_cpu_baseline = 1, True
    if args.roots != "unity":
        args.no_cpu_baseline = True

    import torch
    import zksnark_rs_amd as zk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        args.warmup = max(args.warmup, 1)   # the first collectives (RCCL set-up) never fall into the timed region
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if args.transport == "zk" and args.backend == "gloo":
        args.transport = "zk-gloo"
    use_zk = world > 1 and args.transport in ("zk", "zk-gloo")
    device = local_rank % max(1, torch.cuda.device_count()) if args.backend == "gloo" else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1 and args.transport != "zk":
        import torch.distributed as dist
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))

    ctx = zk.Context(device)
    comm = None
    if use_zk and args.transport == "zk":
        from zksnark_rs_amd.distributed import bootstrap_comm
        comm = bootstrap_comm(ctx, rank, world)     # RCCL communicator inside libzkgpu.so
    elif use_zk:
        from zksnark_rs_amd.distributed import gloo_comm
        comm = gloo_comm(ctx, dist, rank, world)
        store_port = int(os.environ.get("MASTER_PORT", "29500")) + 1
        from datetime import timedelta
        comm._store = torch.distributed.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), store_port, world, rank == 0, timeout=timedelta(seconds=300))

    def barrier():
        if comm is not None:
            comm.barrier()
        elif dist:
            dist.barrier()

    def reduce_max(x):
        if comm is not None:
            return comm.max_f64(float(x))
        if dist:
            t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.backend == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x
    if args.window_bits:
        ctx.set_option("msm_window_bits", args.window_bits)
    if (args.serialize or args.opt) and ctx.get_option("measure_build") != 1:
        raise SystemExit("--serialize / --opt are measurement switches: load the ZK_MEASURE build (make -C zksnark_rs_amd/csrc measure; "
                         "ZKGPU_LIB=zksnark_rs_amd/libzkgpu_measure.so)")
    if args.serialize:
        ctx.set_option("serialize", 1)
    ctx.set_option("msm_shard_points", 1 if args.shard == "points" else 0)
    for kv in args.opt:
        key, val = kv.split("=", 1)
        ctx.set_option(key, int(val))
    inst = build_instance(zk, ctx, args.log_n, args.seed, args.witness, args.roots)
    d_w = torch.from_numpy(inst["weights"].view(np.int64)).cuda()
    m = inst["m"]
    exchange = world > 1 and args.mode == "exchange"
    shard = shard_mode = world > 1 and args.mode == "shard"
    mprover = None
    if exchange and use_zk:
        from zksnark_rs_amd.distributed import MgpuProver
        mprover = MgpuProver(ctx, comm, inst["crs"], inst["qap"])
    elif exchange:
        from zksnark_rs_amd.distributed import GpuExchangeProver, prove_exchange_stream
        xprover = GpuExchangeProver(ctx, inst["crs"], inst["qap"], d_w, m)
    if shard and use_zk:
        from zksnark_rs_amd.distributed import prove_sharded_abi
    elif shard:
        from zksnark_rs_amd.distributed import GpuProver, prove_sharded, prove_sharded_stream
        prover = GpuProver(ctx, inst["crs"], inst["qap"], d_w, m)
        bufs = (prover.new_buffer(zk.PARTIAL_BYTES), prover.new_buffer(world * zk.PARTIAL_BYTES))

    depth = args.depth or (4 if (shard or args.emulate_world > 1) else 2)
    if args.emulate_world and world == 1 and args.mode == "exchange":
        # rank 0's work in rounds of W proofs: one SpMV/NTT stage + W sets of inner products over 1/W of the points;
        # the all-to-alls are local copies of the same size, so the proofs are NOT valid -- timing only
        W = args.emulate_world
        if args.transport == "torch":      # the round-1 Python driver over the stage entry points
            from zksnark_rs_amd.distributed import GpuExchangeProver, prove_exchange_stream
            xp = GpuExchangeProver(ctx, inst["crs"], inst["qap"], d_w, m)
            stream = lambda k: prove_exchange_stream(xp, None, 0, W, [(inst["r"], inst["s"])] * k)   # noqa: E731
        else:                              # the C pipeline (zk_mgpu_push / zk_mgpu_pop) over a loop-back transport
            from zksnark_rs_amd.distributed import Comm, MgpuProver, loopback_comm
            if args.transport == "zk":     # the library's own loop-back (copies on the collectives' stream): stream-ordered hand-overs
                if ctx.get_option("measure_build") != 1:
                    raise SystemExit("--emulate-world --transport zk uses ZK_COMM_LOOPBACK, a measurement switch: load the ZK_MEASURE build "
                                     "(ZKGPU_LIB=zksnark_rs_amd/libzkgpu_measure.so) or pass --transport zk-gloo")
                os.environ["ZK_COMM_LOOPBACK"] = "1"
                lb = Comm(ctx, 0, W, bytes(zk.COMM_ID_BYTES) if hasattr(zk, "COMM_ID_BYTES") else bytes(128))
            else:                          # zk-gloo: a caller's transport (Python callbacks), hand-overs through the host
                lb = loopback_comm(ctx, W)
            mp = MgpuProver(ctx, lb, inst["crs"], inst["qap"])
            stream = lambda k: mp.prove_stream([(d_w.data_ptr(), m, inst["r"], inst["s"])] * k, ahead=2)   # noqa: E731
        for _ in stream(args.warmup):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in stream(args.steps):
            pass
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        free_b, total_b = torch.cuda.mem_get_info()
        print(json.dumps({"diagnostic": "rank 0 of a %d-way scalar-exchange prover (local copies instead of the all-to-alls; %s)"
                                        % (W, "stream-ordered hand-overs" if args.transport == "zk" else "hand-overs through the host"),
                          "ms_per_round_per_rank": round(dt * 1e3, 3), "implied_proofs_per_s_at_%d_gpus" % W: round(W / dt, 2),
                          "hbm_in_use_GiB": round((total_b - free_b) / 2**30, 2)}))
        return
    if args.emulate_world and world == 1:
        from zksnark_rs_amd.distributed import GpuProver, prove_sharded_stream
        prover = GpuProver(ctx, inst["crs"], inst["qap"], d_w, m)
        import types
        W = args.emulate_world
        orig = prover.partial_submit
        prover.partial_submit = lambda rank, world_, r, s, out: orig(0, W, r, s, out)   # rank 0 of W
        prover.combine = lambda gathered, world_, r, s: b""                               # no collective, no assembly
        t0 = time.perf_counter()
        for _ in prove_sharded_stream(prover, None, 0, 1, [(inst["r"], inst["s"])] * args.warmup, depth):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in prove_sharded_stream(prover, None, 0, 1, [(inst["r"], inst["s"])] * args.steps, depth):
            pass
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({"diagnostic": "rank 0 of a %d-way %s-sharded prover, partial sums only" % (W, "point-range" if args.shard == "points" else "window"), "ms_per_proof_per_rank": round(dt * 1e3, 3),
                          "implied_proofs_per_s_at_%d_gpus" % W: round(1.0 / dt, 2)}))
        return

    state = {"degraded": None}
    host_w = None
    if args.witness_from != "hbm" and world == 1:
        host_w = ctx.host_alloc(inst["weights"].shape) if args.witness_from == "pinned" else np.empty_like(inst["weights"])
        host_w[...] = inst["weights"]

    def run(k, local=False):
        """k steps, all submitted and completed inside this call; returns the proof bytes (local: independent provers)."""
        local = local or state["degraded"] is not None
        shard = shard_mode and not local
        if exchange and not local:
            if os.environ.get("ZK_BENCH_TEST_FAIL_EXCHANGE"):   # tests/test_gpu_bench.py: exercises the fallback below
                raise RuntimeError("injected failure of the exchange protocol")
            if mprover is not None:     # zk_mgpu_push / zk_mgpu_pop, two rounds pushed ahead of every pop
                return list(mprover.prove_stream([(d_w.data_ptr(), m, inst["r"], inst["s"])] * k, ahead=2))
            return list(prove_exchange_stream(xprover, dist, rank, world, [(inst["r"], inst["s"])] * k))   # k rounds = k * world proofs
        if shard and use_zk:
            return [prove_sharded_abi(ctx, comm, inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"]) for _ in range(k)]
        if shard and depth == 1:
            return [prove_sharded(prover, dist, rank, world, inst["r"], inst["s"], bufs) for _ in range(k)]
        if shard:
            return list(prove_sharded_stream(prover, dist, rank, world, [(inst["r"], inst["s"])] * k, depth))
        if args.batch > 1:
            out, inflight, left = [], [], k
            while left or inflight:
                if left and len(inflight) < 2:
                    g = min(args.batch, left)
                    inflight.append((ctx.prove_batch_submit(inst["crs"], inst["qap"], [d_w.data_ptr()] * g, [m] * g, [inst["r"]] * g, [inst["s"]] * g), g))
                    left -= g
                else:
                    t, g = inflight.pop(0)
                    out.extend(ctx.prove_batch_wait(t, g))
            return out
        if host_w is not None:
            out, inflight = [], []
            for _ in range(k):
                if len(inflight) == depth:
                    out.append(ctx.prove_wait(inflight.pop(0)))
                inflight.append(ctx.prove_submit_host(inst["crs"], inst["qap"], host_w.ctypes.data, m, inst["r"], inst["s"]))
            while inflight:
                out.append(ctx.prove_wait(inflight.pop(0)))
            return out
        if depth == 1:
            return [ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"]) for _ in range(k)]
        out, inflight = [], []
        for _ in range(k):
            if len(inflight) == depth:
                out.append(ctx.prove_wait(inflight.pop(0)))
            t_s = time.perf_counter()
            inflight.append(ctx.prove_submit(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"]))
            state["submit_s"] = state.get("submit_s", 0.0) + time.perf_counter() - t_s
            state["submits"] = state.get("submits", 0) + 1
        while inflight:
            out.append(ctx.prove_wait(inflight.pop(0)))
        return out

    proof = None
    # The collectives of the sharded protocols have only ever run over gloo before the driver's multi-GPU runs.  If the
    # warm-up fails on any rank (a collective RCCL refuses), every rank falls back to independent provers and the
    # line says so ("degraded") instead of the job producing no number at all.
    err = None
    # Before the W warm-up steps: PRIME untimed steps of one-off setup -- the window tables are built by the first proof, and every
    # proof slot / exchange buffer set the steady state uses is allocated the first time it is touched (three rounds are in flight in
    # the exchange pipeline, so W = 2 alone would leave first-use allocations inside the timed region at N > 1).
    PRIME = 4
    try:
        for p in run(args.warmup + PRIME):
            proof = p
    except Exception as e:   # noqa: BLE001 -- reported in the JSON line
        if world == 1:
            raise
        err = "%s: %s" % (type(e).__name__, str(e)[:300])
    if world > 1:
        if comm is not None:
            # agreement through the bootstrap store (a failed collective may have left the communicator unusable)
            comm._store.set("warm_%d" % rank, b"1" if err else b"0")
            failed = any(bytes(comm._store.get("warm_%d" % g)) == b"1" for g in range(world))
        else:
            flag = torch.tensor([1 if err else 0], dtype=torch.int32, device="cpu" if args.backend == "gloo" else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = bool(int(flag.item()))
        if failed:
            state["degraded"] = err or "another rank failed in the warm-up of --mode %s" % args.mode
            torch.cuda.synchronize()
            for t in range(zk.MAX_IN_FLIGHT):   # tickets the failed protocol left in flight
                try:
                    ctx.prove_wait(t, partial=True)
                except zk.ZkError:
                    pass
            for p in run(args.warmup + PRIME):
                proof = p
    ctx.set_option("profile", 0 if (args.latency or args.no_profile) else 2 if args.kernel_times else 1)
    state["submit_s"], state["submits"] = 0.0, 0
    ctx.profile_reset()
    degraded_now = state["degraded"] is not None    # a failed communicator is not used again: barrier through the bootstrap store

    def store_barrier(tag):
        comm._store.set("%s_%d" % (tag, rank), b"1")
        for g in range(world):
            comm._store.get("%s_%d" % (tag, g))
    torch.cuda.synchronize()
    if degraded_now and comm is not None:
        store_barrier("b0")
    else:
        barrier()
    t0 = time.perf_counter()
    proofs_out = run(args.steps)
    torch.cuda.synchronize()
    if degraded_now and comm is not None:
        store_barrier("b1")
    else:
        barrier()
    elapsed = time.perf_counter() - t0
    if degraded_now and comm is not None:
        comm._store.set("t_%d" % rank, repr(elapsed).encode())
        elapsed = max(float(bytes(comm._store.get("t_%d" % g)).decode()) for g in range(world))
    else:
        elapsed = reduce_max(elapsed)
    prof = ctx.profile()
    ctx.set_option("profile", 0)
    # beside the window-sharded line (north_star, configs[4]): the same K steps as independent provers, one
    # per GPU, no collective -- the throughput mode.  Reported as a secondary object, never as `value`.
    replicas = None
    if state["degraded"] is not None:
        shard = exchange = False
    if shard or exchange:
        run(args.warmup, local=True)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        rep_out = run(args.steps, local=True)
        torch.cuda.synchronize()
        barrier()
        e2 = reduce_max(time.perf_counter() - t1)
        assert all(p == proofs_out[0] for p in rep_out), "replica proof differs from the sharded proof"
        replicas = {"mode": "replicas x%d (independent provers, no collective)" % world, "value": round(world * args.steps / e2, 4),
                    "unit": "proofs/s", "scaling": "weak", "ms_per_step": round(1e3 * e2 / args.steps, 3)}
    for p in proofs_out:
        assert proof is None or p == proof, "non-deterministic proof bytes"
        proof = p
    # beside the resident-witness line: the same steps with every proof's witness handed over in page-locked HOST memory, as the
    # reference's prove(&[T]) does (mod.rs:213-217) -- 96 MB over PCIe per proof at 2^20.  Secondary object, never `value`.
    pcie = None
    if world == 1 and host_w is None and args.batch <= 1 and not shard_mode and not args.latency:
        host_w = ctx.host_alloc(inst["weights"].shape)
        host_w[...] = inst["weights"]
        k2 = max(min(args.steps, 40), 1)
        run(min(args.warmup, 4) or 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out2 = run(k2)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        assert all(p == proof for p in out2), "proof from a host-memory witness differs"
        pcie = {"value": round(k2 / e2, 4), "unit": "proofs/s", "witness_from": "pinned host memory (zk_prove_submit_host)", "steps": k2,
                "host_to_device_MB_per_proof": round(inst["weights"].nbytes / 1e6, 1)}
        ctx.host_free(host_w)
        host_w = None

    proofs = args.steps * (world if (world > 1 and not shard) else 1)   # exchange / replicas: a step is `world` proofs
    value = proofs / elapsed
    if rank == 0:
        total_kernel_ms = sum(e["total_ms"] for e in prof.values()) or 1.0
        # dominant kernel = the bucket accumulation (the only kernels that fill the chip for milliseconds;
        # the event-timed durations of the small reduction kernels include time spent queued behind them)
        acc = {k: v for k, v in prof.items() if k.startswith("msm_accumulate")}
        dom = max(acc.items(), key=lambda kv: kv[1]["total_ms"]) if acc else None
        roofline = None
        n = inst["n"]
        if dom:
            name, e = dom
            g2 = name.endswith("g2")
            avg_ms = e["total_ms"] / e["launches"]
            bytes_per_launch = e["algo_bytes"] / e["launches"]          # 96 B (G1) / 160 B (G2) per (scalar, point) pair
            pairs = bytes_per_launch / (160.0 if g2 else 96.0)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            # additions per launch = pairs x windows, windows averaged over the launches of this kernel in a proof (their point counts
            # differ: L has m - l - 1 points, A n, H + r B1 + s A 2n; the window follows the count; at N > 1 a rank holds 1 / N of each)
            div = world if (world > 1 and not shard) else 1
            counts = [n] if g2 else [inst["m"] - inst["l"] - 1, n, 2 * n - (0 if args.roots == "unity" else 1)]
            wins = [254 // msm_window(max(cnt // div, 1), args.window_bits, g2) + 1 for cnt in counts]
            windows = sum(cnt * w_ for cnt, w_ in zip(counts, wins)) / float(sum(counts))
            gathered = pairs * windows * (4.0 + (128.0 if g2 else 64.0)) + pairs * windows / 32.0 * (304.0 if g2 else 160.0)
            traffic = pmc_traffic(name, args.log_n) if (args.log_n in PMC_FILES and world == 1 and not args.window_bits and args.batch <= 1
                                                       and args.roots == "unity") else None
            adds = pairs * windows
            measured = pmc_acc(name, args.log_n) if args.log_n in PMC_ACC_FILES else None
            valu = None
            if measured:
                winst, slow, src = measured
                g_inst = adds / (avg_ms * 1e-3) / 64.0 * winst / 1e9
                peak_mix = 1.0 / (slow / VALU_PEAK_G["slow"] + (1.0 - slow) / VALU_PEAK_G["fast"])
                meas_mix = 1.0 / (slow / VALU_MEASURED_G["slow"] + (1.0 - slow) / VALU_MEASURED_G["fast"])
                valu = {"achieved": round(g_inst, 1), "unit": "G wave-instr/s", "peak": round(peak_mix, 1), "frac": round(g_inst / peak_mix, 3),
                        "peak_measured_ubench": round(meas_mix, 1), "frac_of_measured_peak": round(g_inst / meas_mix, 3),
                        "additions_per_launch": round(adds), "windows_per_product": wins, "G_additions_per_s": round(adds / (avg_ms * 1e-3) / 1e9, 2),
                        "wave_instr_per_addition": winst, "share_4_cycle_class": slow,
                        "int64_share": src.get("int64_share"), "int32_share": src.get("int32_share"),
                        "sustained_clock_GHz_stand_alone": src.get("sustained_clock_GHz"),
                        "source": "profiles/%s (SQ_INSTS_VALU / additions; SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU)" % PMC_ACC_FILES[args.log_n],
                        "note": "peak = 1024 SIMDs x 2.4 GHz / (share x 4 + (1 - share) x 2 cycles); peak_measured = the same mix at the rates "
                                "tools/ubench_valu.hip sustains (585 / 1062 G/s).  Under this kernel the chip clocks below 2.4 GHz "
                                "(sustained_clock_GHz_stand_alone), i.e. at the sustained clock the issue fraction is frac x 2.4 / clock; other "
                                "kernels of the pipeline share the SIMDs during this measurement"}
            roofline = {
                # the roofline the metric names: SURVEY 8(d) algorithmic bytes of one launch / its event-timed duration against HBM
                "bound": "valu", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_over_algorithmic": round(traffic / bytes_per_launch, 2) if traffic else None,
                "avg_launch_ms": round(avg_ms, 4), "algo_bytes_per_launch": bytes_per_launch, "pairs_per_launch": round(pairs),
                "launches": e["launches"],
                **({"share_of_kernel_time": round(e["total_ms"] / total_kernel_ms, 3)} if args.kernel_times else {}),
                "implementation_bytes_per_launch": round(gathered),
                "note": "frac is the HBM fraction SURVEY 8(d) defines (96 B per scalar-point pair in G1, 160 B in G2).  The kernel is not "
                        "HBM-bound: it is bound by VALU issue (`valu` below).  It gathers one table entry per Pippenger window and pair "
                        "(implementation_bytes = windows x (4 + 64|128) B x pairs + parked images), which is what `traffic` measures.",
                # the bound that binds: wave-instructions issued per second against the issue ceiling of the instruction mix
                "valu": valu,
                "whole_proof": {"algorithmic_bytes": 1404.0 * n, "achieved_GBps": round(1404.0 * n * value / 1e9, 2),
                                "frac": round(1404.0 * n * value / 1e9 / HBM_PEAK_GBS / max(world, 1), 5),
                                "note": "SURVEY 8(d): 1404 n bytes per proof; per GPU"},
            }
        out = {
            "metric": "Groth16 proofs/sec, 2^%d-constraint QAP" % args.log_n,
            "value": round(value, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "u256 (8x u32 Montgomery limbs)",
            "data": "synthetic (chain circuit, %s inputs, SplitMix64 seed %d)" % (args.witness, args.seed),
            "config": {"workload": "synthetic 2^%d-constraint chain QAP (m=%d wires, l=2), BN254, prove() with CRS/QAP/witness resident in HBM%s"
                                   % (args.log_n, m, "" if args.roots == "unity" else "; QAP over the integer roots 1..n" if args.roots == "integers" else "; QAP over caller-supplied (arbitrary) roots"),
                       "parallelism": ("msm-%s-shard x%d + RCCL all-gather" % ("point-range" if args.shard == "points" else "window", world)) if shard
                                      else ("msm point-range shard x%d, NTT stage by proof owner, RCCL all-to-all of scalars and partial sums (%s); "
                                            "a step = one round of %d proofs" % (world, "zk_comm / zk_mgpu inside libzkgpu.so" if use_zk else "torch.distributed", world)) if exchange
                                      else ("replicas x%d" % world),
                       "north_star_deviations": "MSM = fixed-base Pippenger over precomputed window tables with ONE shared set of 2^(c-1) buckets; bucket "
                                                "sums live in registers / HBM images, not LDS (2^16 XYZZ buckets = 9.4 MB against 160 KB); balanced lanes "
                                                "instead of one wavefront per window (DESIGN 4c).  NTT tiles exchange through LDS, not wave shuffles (a 254-bit "
                                                "element is 9 dwords).  N > 1 default = scalar exchange by point ranges; the window-sharded form is --mode shard",
                       "setup_steps_before_warmup": PRIME, "witness_from": args.witness_from, "proofs_in_flight": depth if args.batch <= 1 else "2 batches of %d" % args.batch, "msm_window_bits": args.window_bits or "auto", "proof_sha": __import__("hashlib").sha256(proof).hexdigest()[:16]},
            "roofline": roofline,
            **({"replicas": replicas} if replicas else {}),
            **({"pcie_inclusive": pcie} if pcie else {}),
            **({"degraded": "fell back to independent provers: " + state["degraded"]} if state["degraded"] is not None else {}),
            "hbm_algorithmic_GBps_whole_proof": round(1404.0 * n * value / 1e9, 2),
            "kernel_ms_per_proof": {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(prof.items())},
            **({"host": {"submit_ms_per_proof": round(1e3 * state["submit_s"] / state["submits"], 3),
                         "note": "wall time of zk_prove_submit on the host (everything of a proof is enqueued inside it)"}} if state.get("submits") else {}),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(zk, ctx, args.seed, inst, args.cpu_baseline == "full")
            out["cpu_baseline"]["cores_on_host"] = os.cpu_count()
        print(json.dumps(out))
    if mprover is not None and state["degraded"] is None:
        mprover.close()
    if comm is not None:
        comm.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
