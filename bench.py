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
the NTT stage, one all-gather; scaling "strong"; --shard windows | buckets | points says what a rank owns of every inner product);
--mode replicas runs one independent prover per GPU.  Whatever the primary mode, the N > 1 line also carries the other legs -- and
`config5` (BASELINE config 5: one proof at a time over N GPUs, the faster of the window shard and its balanced bucket-range form).

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


WITNESS_SETS = 4   # distinct (witness, r, s) the timed region cycles through (the sort and the bucket occupancy are data-dependent)


def build_instance(zk, ctx, log_n, seed, witness="uniform", roots="unity", sets=1):
    """The metric's circuit with `sets` distinct (witness, r, s).  Set 0 is drawn exactly as in rounds 1-3 (same generator, same
    order), so its proof -- config.proof_sha -- is comparable across rounds; set j > 0 comes from its own generator."""
    from zksnark_rs_amd.circuits import chain_rows, chain_weights
    n = 1 << log_n
    m, l, u, v, w = chain_rows(log_n)

    def draw(rng):
        x = rng.fr()
        if witness == "boolean":    # inputs a_k in {0, 1}: half of all wires are bits (one very heavy MSM bucket)
            avals = [rng.next() & 1 for _ in range(n)]
        elif witness == "small":    # 32-bit inputs
            avals = [rng.next() & 0xFFFFFFFF for _ in range(n)]
        else:
            avals = [rng.fr() for _ in range(n)]
        return chain_weights(log_n, x, avals)

    rng = zk.SplitMix64(seed)
    weights = draw(rng)
    td = zk.ints_to_limbs([rng.fr() for _ in range(5)])
    r, s = rng.fr(), rng.fr()
    wsets = [dict(weights=weights, r=r, s=s)]
    for j in range(1, sets):
        g = zk.SplitMix64(seed + 1000003 * j)
        wj = draw(g)
        wsets.append(dict(weights=wj, r=g.fr(), s=g.fr()))
    # the same rows over the roots w^j (the metric's workload) or over ASTParser's roots 1..n
    if roots == "arbitrary":    # affine images a k + b of the integers, handed over as caller data: treated as arbitrary field elements
        a, b = rng.fr() | 1, rng.fr()
        k = np.arange(1, n + 1, dtype=object)
        qap = ctx.qap_sparse_roots(zk.ints_to_limbs([int(v_) for v_ in (a * k + b) % zk.R_MODULUS]).reshape(n, 4), m, l, u, v, w)
    else:
        qap = ctx.qap_sparse(log_n, m, l, u, v, w) if roots == "unity" else ctx.qap_sparse_integers(n, m, l, u, v, w)
    crs = ctx.setup(qap, td)      # groth16::setup on the GPU, outside the timed region
    return dict(n=n, m=m, l=l, rows=(u, v, w), qap=qap, crs=crs, weights=weights, td=td, r=r, s=s, log_n=log_n, sets=wsets)


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
# counter passes of THIS round's build (tools/profile_round.sh); a missing file makes the fields null, nothing is typed in here
def _round_file(name, rounds=(6, 5, 4, 3, 2)):
    """the newest committed profiles/rN_<name> (this round's, else the one before it: the line names the file it read)"""
    for r in rounds:
        f = "r%d_%s" % (r, name)
        if os.path.exists(os.path.join(ROOT, "profiles", f)):
            return f
    return "r%d_%s" % (rounds[0], name)


PMC_FILES = {20: _round_file("pmc_traffic.json"), 16: _round_file("pmc_traffic_2p16.json")}   # passes exist for the metric's size and for config 3 (2^16)
PMC_ACC_FILES = {20: _round_file("pmc_acc.json"), 16: _round_file("pmc_acc_2p16.json")}
UBENCH_FILE = _round_file("ubench_valu.txt")   # tools/ubench_valu.hip: sustained issue rates per instruction class, waves per SIMD and chains

# VALU issue ceiling of gfx950 for the two instruction classes of the multiplier:
#   64-bit / integer-multiply class (v_mad_u64_u32, v_mad_i64_i32, v_mul_lo_u32, v_lshl_add_u64, v_ashrrev_i64, carry adds): 4 cycles
#   per wave-instruction and SIMD -> 1024 SIMDs x 2.4 GHz / 4 = 614 G/s; plain 32-bit class (v_and, v_add_u32, v_mov, shifts): 2 cycles.
# What the hardware SUSTAINS of that (585 / 1062 G/s at 8 waves per SIMD) is read from the micro-benchmark's committed output.
VALU_PEAK_G = {"slow": 1024 * 2.4 / 4.0, "fast": 1024 * 2.4 / 2.0}


def ubench_sustained():
    """{"slow": G/s, "fast": G/s} = the best sustained rate of v_mad_u64_u32 / v_add_u32 over the runs of tools/ubench_valu.hip
    (profiles/r2_ubench_valu.txt: op, chains, waves per SIMD, cycles, MHz, G wave-instructions per second, ms); None if missing."""
    try:
        best = {}
        with open(os.path.join(ROOT, "profiles", UBENCH_FILE)) as f:
            for line in f:
                p = line.split()
                if len(p) >= 6 and not line.startswith("#"):
                    best[p[0]] = max(best.get(p[0], 0.0), float(p[5]))
        return {"slow": best["mad_u64_u32"], "fast": best["add_u32"]}
    except Exception:
        return None


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
    if count + 8 >= (1 << 20):
        return 20
    lg = max(count, 1).bit_length() - 1
    return 17 if lg >= 17 else 16 if lg >= 16 else 15 if lg >= 14 else 13 if lg >= 11 else 8


def pmc_file(log_n):
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILES[log_n])) as f:
            return json.load(f)
    except Exception:
        return None


def pmc_traffic(name, log_n=20):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this round's build (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this same command, tools/pmc_summary.py -> profiles/r4_pmc_traffic.json; FETCH_SIZE corrected
    as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be collected inside this process: null when the file is missing."""
    d = pmc_file(log_n)
    try:
        return d["kernels"][PMC_KERNEL[name]]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def pmc_whole_proof(log_n=20):
    """(HBM bytes per proof summed over every kernel of a proof, commit of the build the passes were taken on) or (None, None)"""
    d = pmc_file(log_n)
    if not d or "hbm_bytes_per_proof_corrected" not in d:
        return None, (d or {}).get("commit")
    return d["hbm_bytes_per_proof_corrected"], d.get("commit")


def bounded(fn, seconds, what, device=None):
    """fn() in a worker thread, at most `seconds`: a rendezvous, a communicator set-up or a collective that never completes must end
    in the degraded path, not in a hung job.  The worker is a daemon: if it never returns the process still exits."""
    import threading
    box = {}

    def work():
        try:
            if device is not None:       # the current device is per thread
                import torch
                torch.cuda.set_device(device)
            box["v"] = fn()
        except BaseException as e:   # noqa: BLE001 -- re-raised in the caller
            box["e"] = e
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        raise TimeoutError("%s did not complete within %d s" % (what, seconds))
    if "e" in box:
        raise box["e"]
    return box.get("v")


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
    ap.add_argument("--shard", choices=["windows", "points", "buckets"], default="points",
                    help="what a rank owns of each inner product in --mode shard: buckets = 1 / N of the shared bucket range of every product "
                         "(every window's digits, 1 / N of them kept: entries, accumulation and reduction tail all divide by N); "
                         "Pippenger windows w = rank (mod N), or the "
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
    ap.add_argument("--timeout", type=int, default=180,
                    help="N > 1: bound (s) of every wait on a peer -- rendezvous, communicator set-up, each collective inside the library, "
                         "each leg; what does not complete in time degrades to independent provers")
    ap.add_argument("--sets", type=int, default=WITNESS_SETS, help="distinct (witness, r, s) the timed region cycles through")
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
    device = local_rank % max(1, torch.cuda.device_count())   # one GPU per rank on a multi-GPU node; ranks share a GPU only in the functional runs of the tests
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
    # Every wait on a peer is bounded (--timeout / ZK_BENCH_TIMEOUT_S): the rendezvous and ncclCommInitRank here, the collectives
    # inside the library (ZK_COMM_TIMEOUT_MS -> zk_comm timeouts), each leg of the run below.  What does not complete ends in the
    # `degraded` path (independent provers, said so in the line), never in a hung job.
    TMO = int(os.environ.get("ZK_BENCH_TIMEOUT_S", str(args.timeout)))
    os.environ.setdefault("ZK_COMM_TIMEOUT_MS", str(1000 * TMO))
    boot_err = None
    store = None
    if use_zk:
        from datetime import timedelta
        store_port = int(os.environ.get("MASTER_PORT", "29500")) + 1
        try:
            store = bounded(lambda: torch.distributed.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), store_port, world, rank == 0,
                                                               timeout=timedelta(seconds=TMO)), TMO + 5, "the rendezvous store", device)
        except Exception as e:   # noqa: BLE001
            boot_err = "%s: %s" % (type(e).__name__, str(e)[:200])
    if use_zk and store is not None:
        try:
            if args.transport == "zk":       # RCCL communicator inside libzkgpu.so; the id travels through the store
                from zksnark_rs_amd.distributed import Comm

                def join():
                    if os.environ.get("ZK_BENCH_TEST_HANG_INIT") and rank == world - 1:
                        time.sleep(10 * TMO)
                    if rank == 0:
                        store.set("zk_comm_id", Comm.unique_id())
                    return Comm(ctx, rank, world, bytes(store.get("zk_comm_id")))
                comm = bounded(join, TMO, "zk_comm_init (ncclCommInitRank)", device)
            else:
                from zksnark_rs_amd.distributed import gloo_comm
                comm = gloo_comm(ctx, dist, rank, world)
            comm._store = store
        except Exception as e:   # noqa: BLE001
            boot_err = "%s: %s" % (type(e).__name__, str(e)[:200])
            comm = None

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
    SHARD_OPT = {"windows": 0, "points": 1, "buckets": 2}
    ctx.set_option("msm_shard_points", SHARD_OPT[args.shard])
    for kv in args.opt:
        key, val = kv.split("=", 1)
        ctx.set_option(key, int(val))
    single_set = world > 1 and args.transport == "torch"      # the round-1 Python drivers are bound to one witness
    nsets = 1 if (single_set or args.emulate_world) else max(1, args.sets)
    inst = build_instance(zk, ctx, args.log_n, args.seed, args.witness, args.roots, nsets)
    sets = inst["sets"]
    d_ws = [torch.from_numpy(st["weights"].view(np.int64)).cuda() for st in sets]
    d_w = d_ws[0]
    m = inst["m"]
    exchange = world > 1 and args.mode == "exchange"
    shard = shard_mode = world > 1 and args.mode == "shard"
    mprover = None
    if world > 1 and not use_zk:
        from zksnark_rs_amd.distributed import GpuExchangeProver, GpuProver, prove_exchange_stream, prove_sharded, prove_sharded_stream
        if exchange:
            xprover = GpuExchangeProver(ctx, inst["crs"], inst["qap"], d_w, m)
        if shard:
            prover = GpuProver(ctx, inst["crs"], inst["qap"], d_w, m)
            bufs = (prover.new_buffer(zk.PARTIAL_BYTES), prover.new_buffer(world * zk.PARTIAL_BYTES))
    if use_zk:
        from zksnark_rs_amd.distributed import MgpuProver, prove_sharded_abi

    depth = args.depth or (4 if ((shard and not use_zk) or args.emulate_world > 1) else 2)
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
        print(json.dumps({"diagnostic": "rank 0 of a %d-way %s-sharded prover, partial sums only" % (W, {"points": "point-range", "windows": "window", "buckets": "bucket-range"}[args.shard]), "ms_per_proof_per_rank": round(dt * 1e3, 3),
                          "implied_proofs_per_s_at_%d_gpus" % W: round(1.0 / dt, 2)}))
        return


    state = {"degraded": boot_err, "submit_s": 0.0, "submits": 0}
    host_ws = None
    if args.witness_from != "hbm" and world == 1:
        host_ws = []
        for st in sets:
            hw = ctx.host_alloc(st["weights"].shape) if args.witness_from == "pinned" else np.empty_like(st["weights"])
            hw[...] = st["weights"]
            host_ws.append(hw)

    def job(i, shared=False):
        """the i-th job of a run on this rank: which (witness, r, s).  shared: every rank proves the SAME proof (the latency form)"""
        return (i if shared else i + rank) % len(sets)

    def run(k, mode):
        """k steps in `mode` ("single" = this GPU alone: N = 1 and the replicas leg), all submitted and completed inside the call;
        returns [(set index, proof bytes)]"""
        if mode == "exchange":
            if os.environ.get("ZK_BENCH_TEST_FAIL_EXCHANGE"):   # tests/test_gpu_bench.py: exercises the fallback below
                raise RuntimeError("injected failure of the exchange protocol")
            if os.environ.get("ZK_BENCH_TEST_HANG_EXCHANGE") and rank == world - 1:
                time.sleep(100 * TMO)                             # ... and a peer that never arrives
            idx = [job(i) for i in range(k)]
            if use_zk:      # zk_mgpu_push / zk_mgpu_pop, two rounds pushed ahead of every pop
                return list(zip(idx, state["mprover"].prove_stream([(d_ws[j].data_ptr(), m, sets[j]["r"], sets[j]["s"]) for j in idx], ahead=2)))
            return list(zip(idx, prove_exchange_stream(xprover, dist, rank, world, [(inst["r"], inst["s"])] * k)))   # k rounds = k * world proofs
        if mode in ("shard", "window_shard", "bucket_shard"):
            idx = [job(i, shared=True) for i in range(k)]
            if use_zk:
                return [(j, prove_sharded_abi(ctx, comm, inst["crs"], inst["qap"], d_ws[j].data_ptr(), m, sets[j]["r"], sets[j]["s"])) for j in idx]
            if depth == 1:
                return [(0, prove_sharded(prover, dist, rank, world, inst["r"], inst["s"], bufs)) for _ in range(k)]
            return [(0, pr) for pr in prove_sharded_stream(prover, dist, rank, world, [(inst["r"], inst["s"])] * k, depth)]
        idx = [job(i) for i in range(k)]
        if args.batch > 1:
            out, inflight, left, at = [], [], k, 0
            while left or inflight:
                if left and len(inflight) < 2:
                    g = min(args.batch, left)
                    js = idx[at:at + g]
                    inflight.append((ctx.prove_batch_submit(inst["crs"], inst["qap"], [d_ws[j].data_ptr() for j in js], [m] * g,
                                                            [sets[j]["r"] for j in js], [sets[j]["s"] for j in js]), js))
                    left -= g
                    at += g
                else:
                    t, js = inflight.pop(0)
                    out.extend(zip(js, ctx.prove_batch_wait(t, len(js))))
            return out
        if depth == 1 and host_ws is None:
            return [(j, ctx.prove_dev(inst["crs"], inst["qap"], d_ws[j].data_ptr(), m, sets[j]["r"], sets[j]["s"])) for j in idx]
        out, inflight = [], []
        stamps = state.setdefault("stamps", [])
        for j in idx:
            if len(inflight) == depth:
                jj, t = inflight.pop(0)
                out.append((jj, ctx.prove_wait(t)))
                stamps.append(time.perf_counter())          # completion of one proof while the next ones are in flight
            t_s = time.perf_counter()
            if host_ws is not None:
                t = ctx.prove_submit_host(inst["crs"], inst["qap"], host_ws[j].ctypes.data, m, sets[j]["r"], sets[j]["s"])
            else:
                t = ctx.prove_submit(inst["crs"], inst["qap"], d_ws[j].data_ptr(), m, sets[j]["r"], sets[j]["s"])
            state["submit_s"] += time.perf_counter() - t_s
            state["submits"] += 1
            inflight.append((j, t))
        stamps.append(None)                                  # what follows is the drain of this call: not steady state
        while inflight:
            jj, t = inflight.pop(0)
            out.append((jj, ctx.prove_wait(t)))
        return out

    def completion_intervals_ms():
        """times between consecutive completions of the timed region (diagnostic: ZK_BENCH_DUMP_INTERVALS=1 prints them on stderr)"""
        st = [x for x in state.get("stamps", []) if x is not None]
        return [round(1e3 * (b - a), 3) for a, b in zip(st, st[1:])]

    def steady_state_ms():
        """median time between the completions of consecutive proofs of the timed region, pipeline fill (the first `depth` completions) and
        drain (completions with nothing submitted behind them) excluded; None when too few proofs were timed"""
        st = state.get("stamps", [])
        gaps = [b - a for a, b in zip(st[depth:], st[depth + 1:]) if a is not None and b is not None]
        if len(gaps) < 4:
            return None
        gaps.sort()
        return round(1e3 * gaps[len(gaps) // 2], 4)

    # ---- untimed: one-off set-up and the expected bytes --------------------------------------------------------------------
    # Every set's proof by ONE synchronous single-GPU zk_prove_dev: the bytes every leg of the run (pipelined, batched, from host memory,
    # exchanged, window-sharded, replicated; at N > 1 on every rank) must reproduce.  This also builds the window tables and -- with
    # the PRIME pipelined steps behind it -- touches every proof slot the steady state uses (allocations happen on first use).
    PRIME = 4
    expected = [ctx.prove_dev(inst["crs"], inst["qap"], d_ws[j].data_ptr(), m, sets[j]["r"], sets[j]["s"]) for j in range(len(sets))]
    sha = lambda b: __import__("hashlib").sha256(b).hexdigest()[:16]   # noqa: E731
    # groth16::verify (mod.rs:299-320) of set 0's proof: the pairing check shares no code with the prover or with the closed form the
    # tests use, and it is the only reader of sum_gamma / [gamma]_2 of this CRS (untimed; 36 ms on the host)
    verified = bool(ctx.verify(inst["crs"], sets[0]["weights"][1:1 + inst["l"]], expected[0]))
    for j, pr in run(args.warmup + PRIME, "single"):
        assert pr == expected[j], "pipelined proof differs from the synchronous one"
    torch.cuda.synchronize()

    def agree(tag, bad):
        """True if ANY rank reports `bad` (through the rendezvous store; a rank that never answers counts as bad)"""
        if world == 1:
            return bool(bad)
        if store is None:
            if not dist:
                return bool(bad)
            flag = torch.tensor([1 if bad else 0], dtype=torch.int32, device="cpu" if args.backend == "gloo" else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            return bool(int(flag.item()))
        try:
            store.set("%s_%d" % (tag, rank), b"1" if bad else b"0")
            return any(bytes(store.get("%s_%d" % (tag, g))) == b"1" for g in range(world))
        except Exception:   # noqa: BLE001 -- the store's own timeout
            return True

    def store_barrier(tag):
        store.set("%s_%d" % (tag, rank), b"1")
        for g in range(world):
            store.get("%s_%d" % (tag, g))

    def leg_barrier(tag, collective):
        if world == 1:
            return
        if collective and comm is not None and state["degraded"] is None:
            comm.barrier()
        elif store is not None:
            store_barrier(tag)
        elif dist:
            dist.barrier()

    def leg_max(tag, x, collective):
        if world == 1:
            return x
        if collective and comm is not None and state["degraded"] is None:
            return comm.max_f64(float(x))
        if store is not None:
            store.set("%s_%d" % (tag, rank), repr(float(x)).encode())
            return max(float(bytes(store.get("%s_%d" % (tag, g))).decode()) for g in range(world))
        return reduce_max(x)

    def timed_leg(mode, tag, profile=False):
        """W untimed + K timed steps of `mode`, bracketed by barrier + device synchronisation on both sides, max over ranks.
        Returns (elapsed seconds, mismatches against the expected bytes, kernel profile)."""
        collective = mode != "single"
        run(args.warmup, mode)
        torch.cuda.synchronize()
        prof_ = None
        if profile:
            ctx.set_option("profile", 0 if (args.latency or args.no_profile) else 2 if args.kernel_times else 1)
            state["submit_s"], state["submits"] = 0.0, 0
            ctx.profile_reset()
        leg_barrier(tag + "_b0", collective)
        state["stamps"] = []
        t0 = time.perf_counter()
        outs = run(args.steps, mode)
        state["t_run_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
        state["t_first_ms"] = round(1e3 * (next((x for x in state["stamps"] if x is not None), t0) - t0), 3)
        torch.cuda.synchronize()
        leg_barrier(tag + "_b1", collective)
        elapsed = leg_max(tag + "_t", time.perf_counter() - t0, collective)
        if profile:
            prof_ = ctx.profile()
            ctx.set_option("profile", 0)
        wrong = sum(1 for j, pr in outs if pr != expected[j])
        return elapsed, wrong, prof_

    def leg_record(name, mode, elapsed, wrong, per_step):
        return {"mode": name, "value": round(per_step * args.steps / elapsed, 4), "unit": "proofs/s", "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                "scaling": "strong" if mode in ("shard", "window_shard", "bucket_shard") else "weak",
                "bytes_equal_to_single_gpu_prove": wrong == 0, **({"mismatches": wrong} if wrong else {})}

    # ---- the legs ------------------------------------------------------------------------------------------------------------
    # N = 1: one leg.  N > 1: the scalar exchange (`value`, --mode exchange), the window-sharded latency form (config 5's wording:
    # MSM windows sharded over the GPUs, one collective of the partial sums) and independent replicas -- all in ONE line, each with
    # its own rate and its byte equality against a single-GPU zk_prove_dev of the same inputs.  A leg that fails or does not complete
    # within the bound is reported as such; if it is the primary one, every rank agrees on that through the store and `value` becomes
    # the replicas' (the line says `degraded`).
    legs = {}
    primary = "single" if world == 1 else ("exchange" if exchange else "shard" if shard else "single")
    order = [primary]
    if world > 1:
        if use_zk:
            order += [x for x in ("exchange", "window_shard", "bucket_shard") if x != primary and not (x == "window_shard" and primary == "shard" and args.shard == "windows")
                      and not (x == "bucket_shard" and primary == "shard" and args.shard == "buckets")]
        if "single" not in order:
            order.append("single")
    prof, elapsed, wrong_primary, raw_elapsed = {}, None, 0, {}
    for mode in order:
        collective = mode != "single"
        if collective and state["degraded"] is not None:
            continue
        err = None
        try:
            if mode == "exchange" and use_zk and state.get("mprover") is None:
                state["mprover"] = mprover = bounded(lambda: MgpuProver(ctx, comm, inst["crs"], inst["qap"]), TMO, "zk_mgpu_create", device)
            if mode in ("shard", "window_shard", "bucket_shard"):
                ctx.set_option("msm_shard_points", 0 if mode == "window_shard" else 2 if mode == "bucket_shard" else SHARD_OPT[args.shard])
            if collective:
                res = bounded(lambda: timed_leg(mode, "leg_" + mode, profile=(mode == primary)), max(TMO, 4 * TMO if args.steps > 200 else TMO), "the %s leg" % mode, device)
            else:
                res = timed_leg(mode, "leg_" + mode, profile=(mode == primary))
            if mode == primary:
                state["ss_ms"] = steady_state_ms()
                if os.environ.get("ZK_BENCH_DUMP_INTERVALS"):
                    print("first_completion_ms", state.get("t_first_ms"), "all_waited_ms", state.get("t_run_ms"), "intervals_ms", completion_intervals_ms(), file=sys.stderr, flush=True)
        except Exception as e:   # noqa: BLE001 -- reported in the JSON line
            if not collective:
                raise
            err = "%s: %s" % (type(e).__name__, str(e)[:300])
            res = None
        if collective:
            failed = agree("ok_" + mode, err is not None)
            if failed:
                err = err or "another rank failed or timed out in the %s leg" % mode
                legs[mode] = {"mode": mode, "error": err}
                # a communicator that failed is not used again: abort it (its kernels leave the GPU), release the tickets it left
                if comm is not None and args.transport == "zk":
                    comm.abort()
                if mode == primary or True:
                    state["degraded"] = err
                try:
                    bounded(torch.cuda.synchronize, 30, "device synchronisation after a failed leg", device)
                except Exception:   # noqa: BLE001
                    pass
                for t in range(zk.MAX_IN_FLIGHT):
                    try:
                        ctx.prove_wait(t, partial=True)
                    except zk.ZkError:
                        pass
                continue
        el, wrong, pf = res
        per_step = world if mode in ("exchange", "single") and world > 1 else 1
        name = {"single": "replicas x%d (independent provers, no collective)" % world if world > 1 else "single GPU",
                "exchange": "scalar exchange by point ranges x%d (zk_mgpu_*: all-to-all of scalars and partial sums)" % world,
                "shard": "msm-%s-shard x%d, one proof at a time (latency form)" % ({"points": "point-range", "windows": "window", "buckets": "bucket-range"}[args.shard], world),
                "bucket_shard": "1 / %d of the shared bucket range of every inner product per GPU (every window's digits, 1 / %d kept), one proof at a time, "
                                "all-gather of the partial sums: config 5 balanced -- entries, accumulation and reduction tail all divide by N" % (world, world),
                "window_shard": "MSM windows w = rank (mod %d) per GPU, one proof at a time, all-gather of the partial sums (BASELINE config 5)" % world}[mode]
        legs[mode] = leg_record(name, mode, el, wrong, per_step)
        raw_elapsed[mode] = el
        if mode == "exchange" and state.get("mprover") is not None:
            # zk_mgpu_destroy gives the inner-product streams their whole chip back (option comm_cu_reserve): the legs behind this one
            # -- the window-sharded form, the replicas -- are not measured with compute units set aside for an exchange they do not run
            state["mprover"].close()
            state["mprover"] = None
        if mode == primary:
            prof, elapsed, wrong_primary = pf or {}, el, wrong
    if primary not in legs or "error" in legs.get(primary, {}):     # degraded: `value` is the replicas leg
        primary = "single"
        if "single" not in legs:
            el, wrong, pf = timed_leg("single", "leg_single", profile=True)
            legs["single"] = leg_record("replicas x%d (independent provers, no collective)" % world, "single", el, wrong, world)
            prof, elapsed, wrong_primary = pf or {}, el, wrong
        else:
            elapsed = raw_elapsed["single"]
        shard = exchange = False
    assert wrong_primary == 0, "proof bytes differ from the synchronous single-GPU proof of the same inputs"
    for k_, v_ in legs.items():
        assert v_.get("bytes_equal_to_single_gpu_prove", True), "the %s leg produced different proof bytes" % k_
    proof = expected[0]

    # beside the resident-witness line: the same steps with every proof's witness handed over in page-locked HOST memory, as the
    # reference's prove(&[T]) does (mod.rs:213-217) -- 67 MB over PCIe per proof at 2^20.  Secondary object, never `value`.
    pcie = None
    if world == 1 and host_ws is None and args.batch <= 1 and not args.latency:
        host_ws = []
        for st in sets:
            hw = ctx.host_alloc(st["weights"].shape)
            hw[...] = st["weights"]
            host_ws.append(hw)
        k2 = max(min(args.steps, 40), 1)
        run(min(args.warmup, 4) or 1, "single")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out2 = run(k2, "single")
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        assert all(pr == expected[j] for j, pr in out2), "proof from a host-memory witness differs"
        pcie = {"value": round(k2 / e2, 4), "unit": "proofs/s", "witness_from": "pinned host memory (zk_prove_submit_host)", "steps": k2,
                "host_to_device_MB_per_proof": round(inst["weights"].nbytes / 1e6, 1)}
        for hw in host_ws:
            ctx.host_free(hw)
        host_ws = None

    shard = primary in ("shard", "window_shard", "bucket_shard")
    exchange = primary == "exchange"
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
            n_l, n_hb = inst["m"] - inst["l"] - 1, 2 * n - (0 if args.roots == "unity" else 1)
            # round 5: L and H + r B1 + s A are ONE product over xi_t | xi | sum_delta in a whole proof (option merge_lh; the scalar
            # exchange and point-range shards keep them apart)
            merged = ctx.get_option("merge_lh") == 1 and not exchange and not (primary == "shard" and args.shard == "points")
            counts = [n] if g2 else ([n, n_l + n_hb] if merged else [n_l, n, n_hb])
            wins = [254 // msm_window(max(cnt // div, 1), args.window_bits, g2) + 1 for cnt in counts]
            windows = sum(cnt * w_ for cnt, w_ in zip(counts, wins)) / float(sum(counts))
            gathered = pairs * windows * (4.0 + (128.0 if g2 else 64.0)) + pairs * windows / 32.0 * (304.0 if g2 else 160.0)
            traffic = pmc_traffic(name, args.log_n) if (args.log_n in PMC_FILES and world == 1 and not args.window_bits and args.batch <= 1
                                                       and args.roots == "unity") else None
            traffic_commit = pmc_whole_proof(args.log_n)[1] if args.log_n in PMC_FILES else None
            adds = pairs * windows
            measured = pmc_acc(name, args.log_n) if args.log_n in PMC_ACC_FILES else None
            valu = None
            if measured:
                winst, slow, src = measured
                g_inst = adds / (avg_ms * 1e-3) / 64.0 * winst / 1e9
                peak_mix = 1.0 / (slow / VALU_PEAK_G["slow"] + (1.0 - slow) / VALU_PEAK_G["fast"])
                sus = ubench_sustained()
                meas_mix = 1.0 / (slow / sus["slow"] + (1.0 - slow) / sus["fast"]) if sus else None
                valu = {"achieved": round(g_inst, 1), "unit": "G wave-instr/s", "peak": round(peak_mix, 1), "frac": round(g_inst / peak_mix, 3),
                        "peak_measured_ubench": round(meas_mix, 1) if meas_mix else None, "frac_of_measured_peak": round(g_inst / meas_mix, 3) if meas_mix else None,
                        "peak_measured_source": "profiles/%s" % UBENCH_FILE,
                        "additions_per_launch": round(adds), "windows_per_product": wins, "G_additions_per_s": round(adds / (avg_ms * 1e-3) / 1e9, 2),
                        "wave_instr_per_addition": winst, "share_4_cycle_class": slow,
                        "int64_share": src.get("int64_share"), "int32_share": src.get("int32_share"),
                        "sustained_clock_GHz_stand_alone": src.get("sustained_clock_GHz"),
                        # the issue ceiling of this mix scaled to the clock the chip holds under the kernel (counter passes, stand-alone)
                        "frac_at_sustained_clock": round(g_inst / (meas_mix * src["sustained_clock_GHz"] / 2.4), 3) if (meas_mix and src.get("sustained_clock_GHz")) else None,
                        "source": "profiles/%s (SQ_INSTS_VALU / additions; SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU)" % PMC_ACC_FILES[args.log_n],
                        "note": "peak = 1024 SIMDs x 2.4 GHz / (share x 4 + (1 - share) x 2 cycles); peak_measured = the same mix at the rates "
                                "tools/ubench_valu.hip sustains (read from peak_measured_source).  Under this kernel the chip clocks below 2.4 GHz "
                                "(sustained_clock_GHz_stand_alone), i.e. at the sustained clock the issue fraction is frac x 2.4 / clock; other "
                                "kernels of the pipeline share the SIMDs during this measurement"}
            roofline = {
                # the roofline the metric names: SURVEY 8(d) algorithmic bytes of one launch / its event-timed duration against HBM
                "bound": "valu", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": "profiles/%s" % PMC_FILES[args.log_n] if traffic else None, "traffic_source_commit": traffic_commit if traffic else None,
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
        hbm_proof, hbm_commit = pmc_whole_proof(args.log_n) if (args.log_n in PMC_FILES and args.roots == "unity" and not args.window_bits and args.batch <= 1) else (None, None)
        xg = None
        if world > 1 and use_zk and "exchange" in legs and "error" not in legs["exchange"]:
            el4 = [int(x_) for x_ in ctx.prove_exchange_elems(inst["qap"], world)]
            if el4:
                xg = int(sum(el4) * 32 * (world - 1) // world + zk.PARTIAL_BYTES * (world - 1))
        # the three kernels that carry a proof, each with SURVEY 8(d)'s algorithmic bytes, the counter bytes of the committed PMC passes
        # and their ratio (what DESIGN.md 4 quotes; per PROOF, launches summed): the accumulations gather one table entry per window
        kernels = None
        pf_ = pmc_file(args.log_n) if (args.log_n in PMC_FILES and world == 1 and args.roots == "unity" and not args.window_bits and args.batch <= 1) else None
        if pf_ and pf_.get("kernels") and pf_.get("proofs_sampled"):
            def _per_proof(prefixes):
                tot, launches = 0.0, 0
                for kname, kv in pf_["kernels"].items():
                    if any(kname.startswith(pp) for pp in prefixes):
                        tot += kv["hbm_bytes_per_launch_corrected"] * kv["launches_sampled"]
                        launches += kv["launches_sampled"]
                return tot / pf_["proofs_sampled"], launches / float(pf_["proofs_sampled"])
            rows = (("msm_accumulate_g1", (PMC_KERNEL["msm_accumulate_g1"],), 96.0 * (5 * n - 2), "96 B per scalar-point pair: A (n), H + r B1 + s A (2n - 1), L (m - l - 1 = 2n - 1)"),
                    ("msm_accumulate_g2", (PMC_KERNEL["msm_accumulate_g2"],), 160.0 * n, "160 B per pair: B2 (n)"),
                    ("ntt_tile", ("zk::k_ntt_tile",), 448.0 * n, "7 transforms x 64 B per element"))
            kernels = []
            for kname, prefixes, algo, what in rows:
                cb, lp = _per_proof(prefixes)
                live = prof.get(kname)
                if live and live.get("algo_bytes"):
                    algo = live["algo_bytes"] / float(args.steps)      # the library's own count of the pairs it multiplied
                kernels.append({"kernel": kname, "launches_per_proof": round(lp, 2), "algorithmic_bytes_per_proof": algo, "algorithmic": what,
                                "counter_bytes_per_proof": round(cb), "counter_over_algorithmic": round(cb / algo, 2) if algo else None,
                                "avg_launch_ms_live": round(live["total_ms"] / live["launches"], 4) if live else None,
                                "source": "profiles/%s" % PMC_FILES[args.log_n]})
        out = {
            "metric": "Groth16 proofs/sec, 2^%d-constraint QAP" % args.log_n,
            "value": round(value, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            # median time between consecutive proof completions of the timed region, pipeline fill and drain excluded (what
            # profiles/r5_timeline_pipelined_2p20.txt shows as the period); ms_per_step includes both
            "steady_state_ms_per_proof": state.get("ss_ms"),
            "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "u254 mod p (9x29-bit lazy Montgomery limbs, R=2^261; 8xu32 at rest)",
            "data": "synthetic (chain circuit, %s inputs, SplitMix64 seed %d; %d distinct (witness, r, s) cycled through the timed region)" % (args.witness, args.seed, len(sets)),
            "config": {"workload": "synthetic 2^%d-constraint chain QAP (m=%d wires, l=2), BN254, prove() with CRS/QAP/witness resident in HBM%s"
                                   % (args.log_n, m, "" if args.roots == "unity" else "; QAP over the integer roots 1..n" if args.roots == "integers" else "; QAP over caller-supplied (arbitrary) roots"),
                       "parallelism": ("msm-%s-shard x%d + RCCL all-gather" % ("window" if (primary == "window_shard" or args.shard == "windows") else "point-range", world)) if shard
                                      else ("msm point-range shard x%d, NTT stage by proof owner, RCCL all-to-all of scalars and partial sums (%s); "
                                            "a step = one round of %d proofs" % (world, "zk_comm / zk_mgpu inside libzkgpu.so" if use_zk else "torch.distributed", world)) if exchange
                                      else ("replicas x%d" % world),
                       "north_star_deviations": "MSM = fixed-base Pippenger over precomputed window tables with ONE shared set of 2^(c-1) buckets; bucket "
                                                "sums live in registers / HBM images, not LDS (2^16 XYZZ buckets = 9.4 MB against 160 KB); balanced lanes "
                                                "instead of one wavefront per window (DESIGN 4c).  NTT tiles exchange through LDS, not wave shuffles (a 254-bit "
                                                "element is 9 dwords).  N > 1: `value` = scalar exchange by point ranges; config 5's window-sharded form is the "
                                                "`window_shard` object of the same line",
                       "setup_steps_before_warmup": PRIME + len(sets), "witness_from": args.witness_from, "witness_sets": len(sets),
                       "proofs_in_flight": depth if args.batch <= 1 else "2 batches of %d" % args.batch, "msm_window_bits": args.window_bits or "auto",
                       "proof_sha": sha(expected[0]), "proof_shas": [sha(e_) for e_ in expected], "verified": verified,
                       "expected_bytes_from": "one synchronous single-GPU zk_prove_dev per set in the untimed set-up; every timed proof is compared with it"},
            "roofline": roofline,
            "kernels": kernels,
            **({"config5": (lambda w_, b_: None if not (w_ or b_) else (lambda best_, form_: {
                "value": best_["value"], "ms_per_step": best_["ms_per_step"], "scaling": "strong", "unit": "proofs/s", "form": form_,
                "what": "BASELINE config 5: ONE 2^%d proof at a time with every inner product sharded over the %d GPUs and one RCCL all-gather of the "
                        "768-byte partial sums; the faster of `window_shard` (the literal partition: windows w = rank mod N of the fixed-base tables) "
                        "and `bucket_shard` (the tables share ONE bucket set over all windows: a rank owns 1 / N of it -- equal shares for every N)" % (args.log_n, world),
                "window_shard_value": w_["value"] if w_ else None, "bucket_shard_value": b_["value"] if b_ else None})(
                    *((b_, "bucket_shard") if (b_ and (not w_ or b_["value"] >= w_["value"])) else (w_, "window_shard"))))(
                    legs.get("window_shard") or (legs.get("shard") if args.shard == "windows" else None),
                    legs.get("bucket_shard") or (legs.get("shard") if args.shard == "buckets" else None))} if world > 1 else {}),
            **({k_: v_ for k_, v_ in (("exchange", legs.get("exchange")), ("window_shard", legs.get("window_shard") or (legs.get("shard") if args.shard == "windows" else None)),
                                      ("bucket_shard", legs.get("bucket_shard") or (legs.get("shard") if args.shard == "buckets" else None)),
                                      ("shard", legs.get("shard") if args.shard == "points" else None), ("replicas", legs.get("single"))) if v_ and world > 1}),
            **({"rccl_ranks": comm.rccl_ranks() if (comm is not None and state["degraded"] is None) else 0, "xgmi_bytes_sent_per_rank_per_round": xg,
                "comm_cu_reserve_per_xcd": ctx.get_option("comm_cu_reserve") if (use_zk and args.transport == "zk") else 0, "wait_bound_s": TMO} if world > 1 else {}),
            **({"pcie_inclusive": pcie} if pcie else {}),
            **({"degraded": "fell back to independent provers: " + state["degraded"]} if state["degraded"] is not None else {}),
            "hbm_algorithmic_GBps_whole_proof": round(1404.0 * n * value / 1e9, 2),
            "hbm_measured_GBps_whole_proof": round(hbm_proof * value / max(world, 1) / 1e9, 1) if (hbm_proof and world == 1) else None,
            "hbm_measured_source": {"file": "profiles/%s" % PMC_FILES[args.log_n], "commit": hbm_commit, "bytes_per_proof": hbm_proof,
                                    "note": "sum over every kernel of a proof of (corrected FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 --pmc passes of this command, x value"} if hbm_proof else None,
            "kernel_ms_per_proof": {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(prof.items())},
            **({"host": {"submit_ms_per_proof": round(1e3 * state["submit_s"] / state["submits"], 3),
                         "note": "wall time of zk_prove_submit on the host (everything of a proof is enqueued inside it)"}} if state.get("submits") else {}),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(zk, ctx, args.seed, inst, args.cpu_baseline == "full")
            out["cpu_baseline"]["cores_on_host"] = os.cpu_count()
        print(json.dumps(out), flush=True)
    hung = state["degraded"] is not None and world > 1
    if hung:
        # a communicator that failed or timed out may still hold a blocked worker thread or a blocked peer: leave without the
        # orderly tear-down (which would wait for them)
        sys.stdout.flush()
        os._exit(0)
    if state.get("mprover") is not None:
        state["mprover"].close()
    if comm is not None:
        comm.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
