#!/usr/bin/env python3
"""bench.py -- queries/s of the MI355X RaBitQ engine at recall@10 >= 0.95 on the BASELINE.json
workload (configs[2]: 100M x 128 synthetic, 4096 lists, nprobe 64, one GPU), with the scan kernel's
roofline figures and the CPU baseline (oracle port of the reference's AVX2 path) in the same line.

A "step" is one pass of the hot path (rotate -> coarse rank -> per-list query quantise -> binary
scan -> rerank -> ordered replay) over one batch of synthetic queries that are already resident
in HBM.  N > 1: one process per GPU (torch.distributed / RCCL); every rank indexes its own
100M-vector shard under the shared centroid set, answers the same batch against its shard, and
the per-shard top-k are merged with one all-gather per batch (rabitq_amd/sharding.py).

    python bench.py --gpus 1 --steps 5 --warmup 1
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2,
                    help="untimed steps; two by default: the first one learns the survivor-buffer capacity of the workload, the "
                         "second one runs with it and re-allocates the workspace once (tens of GB on the hard distribution: "
                         "0.5-2 s that would otherwise land in the first timed step)")
    # workload (defaults = BASELINE.json configs[2], the configuration the metric is quoted on)
    ap.add_argument("--vectors", type=int, default=None,
                    help="vectors per GPU (default 100M = BASELINE configs[2] at EVERY --gpus N, so that N = 1, 2, 4, 8 are one curve; "
                         "the 8-GPU run then also measures BASELINE configs[4] -- 125M per GPU = 1B x 128 over 32 768 lists -- into "
                         "`secondary.configs4_1Bx128`)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--lists", type=int, default=4096, help="IVF lists per GPU")
    ap.add_argument("--nprobe", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None,
                    help="queries per step over ALL GPUs (default 65536 x N: the lists grow with N -- 4096 per GPU --, so a fixed "
                         "batch would leave every rank's matrix-core scan N times fewer queries per list; with the batch scaled, a "
                         "rank's (query, list) pairs per step stay what they are on one GPU.  Throughput grows with the batch: "
                         "more queries share each list)")
    ap.add_argument("--sigma", type=float, default=0.5)
    ap.add_argument("--distribution", choices=["easy", "hard"], default="easy",
                    help="easy = SURVEY.md 8(d) mixture with the true centres as centroids; hard = overlapping clusters, "
                         "Zipf list sizes, k-means-trained centroids")
    ap.add_argument("--hard-centre-ratio", type=float, default=1.0, help="hard: centre std / sigma (1 = clusters overlap)")
    ap.add_argument("--zipf", type=float, default=0.7, help="hard: list-size exponent")
    ap.add_argument("--device-base-gb", type=float, default=0.0,
                    help="HBM budget of the raw vectors (0 = automatic); the rest is kept in pinned host memory")
    ap.add_argument("--centre-scale", type=float, default=1.0)
    ap.add_argument("--gt-queries", type=int, default=1000)
    ap.add_argument("--cpu-queries", type=int, default=1000,
                    help="CPU-baseline sample: the 1000-query subset of SURVEY.md 8(d) (0 = skip); bounded to ~25 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--small-batch", default="64,16,1",
                    help="batch sizes of the HBM-regime scan measurement (lists hardly shared; '0' or '' = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo = single-GPU rehearsal)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--sharded-path", action="store_true",
                    help="rehearsal on one GPU: run the multi-GPU step (rq_query_batch_sharded_device: handshake, coarse "
                         "ranking + probe-list all-gather, shared thresholds, top-k all-gather) on an RCCL communicator of one "
                         "rank, to see what the extra plumbing costs")
    ap.add_argument("--collective-path", choices=["c-abi", "torch"], default="c-abi",
                    help="multi-GPU step: c-abi = rq_query_batch_sharded_device over an ncclComm_t created here (what a host "
                         "without torch binds); torch = the same step assembled from the per-call entries and "
                         "torch.distributed collectives (also the in-process fallback if the C-ABI step fails in warm-up)")
    ap.add_argument("--secondary", action=argparse.BooleanOptionalAction, default=True,
                    help="one-GPU runs also measure, in the same process and into the same JSON line (`secondary`), the hard "
                         "distribution and BASELINE configs[3] (100M x 768); --no-secondary skips them")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="engine option for this run (rq_set_option; e.g. stage_growth=16, rerank_shadow=0): experiments only, "
                         "recorded in the output line")
    ap.add_argument("--two-in-flight", action=argparse.BooleanOptionalAction, default=True,
                    help="after the timed region (whose steps run one batch at a time, so that every launch runs alone and "
                         "per-kernel times from HIP events and rocprofv3 --stats stay comparable), also measure the throughput "
                         "with two batches in flight (rq_query_batch_device_begin/_end) and report it as an extra field")
    ap.add_argument("--batch-sweep", action=argparse.BooleanOptionalAction, default=True,
                    help="headline run: also time batches of 128 ... 16384 (and 2x / 4x the headline batch) on the same index, fresh "
                         "queries per call (`batch_sweep`); the profiling passes switch it off so that every launch of the dominant "
                         "kernel in their tables is a full-batch launch")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="one-GPU rehearsal of ONE rank of a W-GPU run: W x --lists centroids of which this process owns the first "
                         "--lists (its vectors come from them), queries drawn over all of them, the C-ABI multi-GPU step with a "
                         "one-rank communicator (shared-threshold passes as on a real rank; the coarse ranking is NOT sliced, "
                         "so its share is W times a real rank's).  Per-kernel times show where a rank's step goes at W GPUs")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="batches kept in flight in the timed loop (rq_query_batch_device_begin/_end); 1 = one blocking "
                         "call per step (default: per-kernel times are then clean); see --two-in-flight")
    args = ap.parse_args()
    explicit_workload = (args.vectors is not None or args.dim != 128 or args.distribution != "easy" or args.emulate_world > 1 or
                         args.batch is not None)
    if args.emulate_world > 1:
        assert args.gpus == 1, "--emulate-world is a one-GPU rehearsal"
        args.sharded_path = True
    if args.vectors is None:
        args.vectors = 100_000_000
    default_batch = args.batch is None
    if args.batch is None:
        args.batch = 65536 * max(1, args.gpus) * max(1, args.emulate_world)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and wait.
        # This parent never imports torch or touches the GPU, and nothing is exec'd from a process that has.
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import rabitq_amd
    from rabitq_amd import _lib, index as rqi, ops, sharding
    from tests import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        keep = os.dup(1)          # some backends (gloo) chat on stdout while connecting: stdout carries the JSON line only
        os.dup2(2, 1)
        try:
            dist.init_process_group(args.backend, rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    if args.same_device:
        local_rank = 0
    assert world == max(args.gpus, 1), f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _lib.check(_lib.lib().rq_init(local_rank))
    from rabitq_amd import index as _ix
    for ov in args.option:
        name, _, val = ov.partition("=")
        _ix.set_option(name, int(val))

    ctx = {"world": world, "rank": rank, "dev": dev}
    line = run_workload(args, ctx, extras=True)
    # ---- the configurations only the builder had run so far, measured in the SAME process into the SAME line ------------
    if world == 1 and not args.sharded_path and args.secondary and not explicit_workload:
        import copy
        sec = {}
        for name, over in (("hard_distribution_100Mx128", {"distribution": "hard"}),
                           ("config3_100Mx768", {"dim": 768, "batch": 32768})):
            a2 = copy.copy(args)
            for key, v in over.items():
                setattr(a2, key, v)
            a2.steps, a2.warmup, a2.gt_queries = min(args.steps, 5), 3, 256   # (the hard distribution sizes its survivor arena over the first three passes)
            try:
                full = run_workload(a2, ctx, extras=False)
                keep = ("value", "unit", "ms_per_step", "steps", "warmup", "config", "recall_at_10", "recall_queries", "build_seconds",
                        "build", "kernel_ms_per_step", "matrix_exact_path_rate", "rerank_candidates_per_query", "rerank_shadow_rejects_per_query", "retries",
                        "rough_per_query", "precise_per_query", "roofline", "roofline_rotation", "survivor_workspace_GB")
                sec[name] = {key: full[key] for key in keep if key in full}
            except Exception as e:   # a secondary failure must not cost the headline line
                import traceback
                traceback.print_exc()
                sec[name] = {"error": f"{type(e).__name__}: {e}"}
        line["secondary"] = sec
    if world == 8 and args.secondary and not explicit_workload:
        # BASELINE configs[4] (1B x 128 over 8 GPUs; k is not given there: 32 768 = 4096 per GPU) next to the
        # 100M-per-GPU point of the scaling curve: 125M vectors per GPU, same lists / batch, same process group
        import copy
        a2 = copy.copy(args)
        a2.vectors, a2.steps, a2.warmup, a2.gt_queries = 125_000_000, min(args.steps, 5), 2, 256
        try:
            full = run_workload(a2, ctx, extras=False)
            keep = ("value", "unit", "ms_per_step", "steps", "warmup", "config", "recall_at_10", "recall_queries", "build_seconds",
                    "kernel_ms_per_step", "collective_path", "roofline", "rough_per_query", "precise_per_query")
            line["secondary"] = {"configs4_1Bx128": {key: full[key] for key in keep if key in full}}
        except Exception as e:   # must not cost the headline line (every rank takes the same path: the failure modes are collective)
            import traceback
            traceback.print_exc()
            line["secondary"] = {"configs4_1Bx128": {"error": f"{type(e).__name__}: {e}"}}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_workload(args, ctx, extras=True):
    """Build the index of one workload, run the timed steps, return the bench line (a dict).  extras: the small-batch /
    single-query / two-in-flight / CPU-baseline legs of the headline run."""
    import torch
    import torch.distributed as dist
    import rabitq_amd
    from rabitq_amd import index as rqi, ops, sharding
    from tests import synth
    world, rank, dev = ctx["world"], ctx["rank"], ctx["dev"]
    n, d, k_local, nprobe, topk, B = args.vectors, args.dim, args.lists, args.nprobe, args.topk, args.batch
    k = k_local * world * max(1, getattr(args, "emulate_world", 1))   # global list count; every rank knows all centroids
    t0 = time.time()

    # ---- synthetic inputs: mixture of Gaussians, generated on device ------------
    # "easy": true centres as centroids, uniform list sizes, centres far apart (every query's neighbours sit in its
    # first list).  "hard": overlapping clusters (centre spacing ~ cluster radius), Zipf list sizes, and centroids
    # TRAINED by rq_kmeans_device on a sample instead of the generating centres.
    hard = args.distribution == "hard"
    centre_scale = args.centre_scale if not hard else args.sigma * args.hard_centre_ratio
    centres = synth.device_centres(k, d, dev, centre_scale)
    weights = None
    if hard:
        wz = 1.0 / torch.arange(1, k_local + 1, device=dev, dtype=torch.float64) ** args.zipf
        weights = (wz / wz.sum()).float()[torch.randperm(k_local, device=dev, generator=torch.Generator(device=dev).manual_seed(5))]
    my_lo = rank * k_local                   # this rank's points come from its own k_local centres
    qweights = None if weights is None else weights.repeat(world) / world
    queries = synth.device_queries(centres, B, args.sigma, dev, weights=qweights)
    # the timed steps rotate over FOUR distinct query batches (set 0 = `queries`: warm-up, ground truth, every extra leg): what the
    # warm-up learns about set 0 -- survivor capacities, the arena's size -- has to hold for batches the engine has never seen
    query_sets = [queries] + [synth.device_queries(centres, B, args.sigma, dev, seed=7 + 1000 * i, weights=qweights) for i in (1, 2, 3)]
    qsel = [0]   # the set the next step answers
    P = synth.random_orthogonal(d, seed=99)
    chunk = max(262_144, min(4_000_000, (512 << 20) // d))     # rows per generated chunk (4M at d = 128)
    chunks = [(ci, i0, min(chunk, n - i0)) for ci, i0 in enumerate(range(0, n, chunk))]

    def gen(ci, i0, m):
        return synth.device_mixture_chunk(centres, i0, m, args.sigma, ci, 42 + 1000 * rank, my_lo, k_local, weights)[0]

    centroids = centres
    kmeans_s = None
    if hard:   # centroid training (scripts/cluster.py's role): Lloyd on a 256-points-per-centroid sample of the same mixture
        tk = time.time()
        ns = min(n, 256 * k_local)
        sample = synth.device_mixture_chunk(centres, 0, ns, args.sigma, 999_983, 42 + 1000 * rank, my_lo, k_local, weights)[0].contiguous()
        learnt = torch.zeros((k_local, d), device=dev, dtype=torch.float32)
        ops.kmeans_device(sample.data_ptr(), ns, d, k_local, learnt.data_ptr(), iters=10, seed=3)
        if world > 1:   # every rank needs all centroids: one all-gather at build time
            allc = [torch.zeros_like(learnt) for _ in range(world)] if args.backend != "gloo" else None
            if allc is None:
                lc = [torch.zeros((k_local, d)) for _ in range(world)]
                dist.all_gather(lc, learnt.cpu())
                centroids = torch.cat(lc).to(dev)
            else:
                dist.all_gather(allc, learnt)
                centroids = torch.cat(allc)
        else:
            centroids = learnt
        del sample
        kmeans_s = time.time() - tk
        log(f"centroids trained on {ns} samples in {kmeans_s:.1f}s")

    # ---- pass 1 of the streamed build + brute-force ground truth, chunk by chunk (the base is never resident as
    # a whole: 100M x 768 would not fit beside its own cluster-ordered copy) ------------------------------------
    # Ground truth in f64 on purpose: the f32 GEMM form |q|^2 - 2 q.x + |x|^2 cancels catastrophically here (neighbour
    # distances differ by ~0.1 under norms of ~160) and an f32 library GEMM gave a wrong ground truth.
    ngt = min(args.gt_queries, B)
    qg = queries[:ngt].double()
    best_d = torch.full((ngt, topk), float("inf"), device=dev, dtype=torch.float64)
    best_i = torch.full((ngt, topk), -1, device=dev, dtype=torch.int64)
    qn = (qg * qg).sum(1, keepdim=True)
    gchunk = max(65_536, min(1_000_000, (128 << 20) // d))
    build_s = 0.0
    builder = rabitq_amd.RaBitQ.builder(n, d, centroids.data_ptr(), k, orthogonal=P,
                                        max_device_base_bytes=int(args.device_base_gb * (1 << 30)))
    for ci, i0, m in chunks:
        xc = gen(ci, i0, m)
        for j0 in range(0, m if ngt else 0, gchunk):
            xb = xc[j0:j0 + gchunk].double()
            d2 = qn - 2.0 * (qg @ xb.T) + (xb * xb).sum(1)[None, :]
            cd, ci_ = torch.topk(d2, min(topk, xb.shape[0]), dim=1, largest=False)
            alld = torch.cat([best_d, cd], 1)
            alli = torch.cat([best_i, ci_ + i0 + j0 + rank * n], 1)
            sel = torch.topk(alld, topk, dim=1, largest=False).indices
            best_d, best_i = torch.gather(alld, 1, sel), torch.gather(alli, 1, sel)
            del d2, xb
        torch.cuda.synchronize()
        tb = time.perf_counter()
        builder.assign_chunk(xc.data_ptr(), i0, m)
        build_s += time.perf_counter() - tb
        del xc
    if world > 1 and ngt:   # global ground truth = merge of the shards' exact top-k
        pay = sharding.pack_topk(best_d.float(), best_i - rank * n, torch.full((ngt,), topk, device=dev), rank * n)
        if args.backend == "gloo":
            pay = pay.cpu()
        _, best_i, _ = sharding.merge_shard_topk(pay, topk, id_bound=world * n)
    gt = best_i.cpu().numpy()
    del best_d, best_i, qg, qn
    torch.cuda.empty_cache()     # the engine sizes the HBM tier of the raw vectors by what is free now
    log(f"pass 1 (rotate, assign, quantise) + ground truth done ({time.time() - t0:.1f}s)")
    tb = time.perf_counter()
    builder.order()
    build_s += time.perf_counter() - tb
    bstats = builder.stats()
    # ---- pass 2: the raw vectors to their cluster-order positions (HBM tier, or pinned host memory beyond it) -----
    for ci, i0, m in chunks:
        xc = gen(ci, i0, m).contiguous()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        builder.place_chunk(xc.data_ptr(), i0, m)
        build_s += time.perf_counter() - tb
        del xc
    tb = time.perf_counter()
    idx = builder.finish()
    build_s += time.perf_counter() - tb
    torch.cuda.empty_cache()
    # rotation (a5) on the matrix cores over the WHOLE build: 2 n dim^2 flop in the rotate kernel's summed device time
    rot_ms = bstats["ms_rotate"]
    tf = 2.0 * n * idx.dim * idx.dim / (rot_ms * 1e-3) / 1e12
    rotation = {"bound": "mfma", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                "kernel": "rotate_mfma_kernel (v_mfma_f32_32x32x2_f32, exact f32)", "rows": n,
                "flops": 2.0 * n * idx.dim * idx.dim, "ms": round(rot_ms, 3),
                "GBps_in_plus_out": round(2 * n * idx.dim * 4 / (rot_ms * 1e-3) / 1e9, 1),
                "note": "every row of the build, HIP events around the kernel on its launch stream, summed over chunks"}
    build_info = {"assign_ms": round(bstats["ms_assign"], 1), "quantize_ms": round(bstats["ms_quantize"], 1),
                  "rotate_ms": round(rot_ms, 1), "assign_rows_redone_in_exact_order": bstats.get("rows_exact_redo"), "rows_in_hbm": idx.n_hbm, "rows_in_pinned_host_memory": idx.n - idx.n_hbm, "raw_vectors_as_split_rows": idx.split_rows,
                  "centroid_training_seconds": None if kmeans_s is None else round(kmeans_s, 1)}
    log(f"index built in {build_s:.1f}s of engine time: n={idx.n} dim={idx.dim} k={idx.k} max_list_len={idx.max_list_len} "
        f"rows in HBM {idx.n_hbm} / host {idx.n - idx.n_hbm}")

    sharded = world > 1 or args.sharded_path
    depth = max(1, args.pipeline) if not sharded else 1
    outs = [(torch.empty((B, topk), device=dev, dtype=torch.float32), torch.zeros((B, topk), device=dev, dtype=torch.int32),
             torch.zeros((B,), device=dev, dtype=torch.int32)) for _ in range(max(depth, 2))]
    out_d, out_i, out_n = outs[0]

    # ---- the multi-GPU step -------------------------------------------------------------------------------------------
    # c-abi: rq_query_batch_sharded_device on an ncclComm_t created here (ncclGetUniqueId on rank 0, the 128 bytes
    # broadcast through the process group, ncclCommInitRank; RCCL = the copy torch carries, which is also what the engine
    # binds with dlsym).  With --backend gloo (one-GPU rehearsal) the same entry runs on host-buffer collectives.
    # torch: the same step from the per-call entries + torch.distributed; also the in-process fallback.
    collective_path = None
    comm_handle, keep_alive = 0, []
    pc_local = pd_local = seeded = None
    extra_prof = []   # torch path: profile of the first engine call of a step (the second one is last_profile())

    def all_agree(ok: bool) -> bool:
        if world == 1:
            return ok
        t = torch.tensor([1.0 if ok else 0.0], device="cpu" if args.backend == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    if sharded and args.collective_path == "c-abi":
        ok, why = True, ""
        try:
            if getattr(args, "emulate_world", 1) > 1:
                hc = sharding.EmulatedPeers()      # one rank of W: the absent peers' threshold seeds are filled in (see the class)
                hc.install()
                keep_alive.append(hc)
                comm_handle = 1
            elif args.backend == "gloo":
                hc = sharding.HostCollectives()
                hc.install()
                keep_alive.append(hc)
                comm_handle = 1          # passed through to the callbacks untouched
            else:
                os.environ.setdefault("RABITQ_RCCL_LIB", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
                # ncclCommInitRank blocks until EVERY rank has called it: first make sure every rank can (library and symbols
                # present), so that a rank which cannot does not leave the others waiting in the bootstrap
                can = True
                try:
                    lib_ = sharding._rccl_lib()
                    lib_.ncclGetUniqueId, lib_.ncclCommInitRank, lib_.ncclAllGather, lib_.ncclAllReduce, lib_.ncclCommUserRank   # noqa: B018
                except Exception as e:
                    can, why = False, f"{type(e).__name__}: {e}"
                if not all_agree(can):
                    raise RuntimeError(why or "RCCL is not available on another rank")
                rc = sharding.RcclComm(rank, world)
                keep_alive.append(rc)
                comm_handle = rc.handle
            if world == 1:
                rqi.set_option("shared_thresholds", 2)   # rehearsal: the shared-threshold step also with one rank
        except Exception as e:
            ok, why = False, f"{type(e).__name__}: {e}"
        if all_agree(ok):
            collective_path = "rq_query_batch_sharded_device/" + ("emulated peers (one-GPU rehearsal of one rank)" if getattr(args, "emulate_world", 1) > 1 else
                                                                  "host-buffer collectives (gloo rehearsal)" if args.backend == "gloo" else "RCCL")
        else:
            log(f"C-ABI collective path unavailable ({why or 'another rank failed'}): falling back to torch.distributed")
            if keep_alive and isinstance(keep_alive[0], sharding.HostCollectives):   # (also an EmulatedPeers table)
                sharding.HostCollectives.uninstall()
    if sharded and collective_path is None:
        collective_path = "per-call entries + torch.distributed (" + args.backend + ")"
        pc_local = torch.zeros((B, nprobe), device=dev, dtype=torch.int32)
        pd_local = torch.zeros((B, nprobe), device=dev, dtype=torch.float32)
        seeded = sharding.SeededShardQuery(B, topk, dev)

    def step_torch():
        # each rank ranks only the lists it owns; one all-gather merges the per-rank nearest lists
        idx.coarse_topk_device(query_sets[qsel[0]].data_ptr(), B, d, rank * k_local, (rank + 1) * k_local, nprobe,
                               pc_local.data_ptr(), pd_local.data_ptr())
        pcl, pdl = (pc_local.cpu(), pd_local.cpu()) if args.backend == "gloo" else (pc_local, pd_local)
        pc, pdist = sharding.merge_probe_lists(pcl, pdl, nprobe)
        pc, pdist = pc.to(dev), pdist.to(dev)
        # thresholds shared between the shards (sharding.SeededShardQuery): the nearest list alone, one
        # all-reduce(min) of the k-th best distances found there, then the other lists seeded with it -- a shard that
        # does not hold a query's neighbourhood would otherwise re-rank most of what it scans
        seeded.run(idx, query_sets[qsel[0]].data_ptr(), d, pc, pdist, cpu_collectives=args.backend == "gloo")
        extra_prof.append(seeded.profile_a)
        pay = seeded.payload(rank * n)
        if args.backend == "gloo":
            pay = pay.cpu()
        return sharding.merge_shard_topk(pay, topk, id_bound=world * n)

    def step_c_abi():
        idx.query_batch_sharded_device(comm_handle, world, rank * n, query_sets[qsel[0]].data_ptr(), B, d, nprobe, topk, out_d.data_ptr(),
                                       out_i.data_ptr(), out_n.data_ptr())
        return out_d, out_i.to(torch.int64) & 0xFFFFFFFF, out_n

    def step():
        if not sharded:
            idx.query_batch_device(query_sets[qsel[0]].data_ptr(), B, d, nprobe, topk, out_d.data_ptr(), out_i.data_ptr(),
                                   out_n.data_ptr())
            return out_d, out_i.to(torch.int64) & 0xFFFFFFFF, out_n
        return step_c_abi() if seeded is None else step_torch()

    if sharded and seeded is None:
        # one untimed step decides: a non-OK status of the C-ABI step on any rank -> every rank switches to the torch path,
        # in this process (the entry reports a failed rank to all of them, so they agree; the all-reduce makes sure)
        ok, why = True, ""
        try:
            step_c_abi()
        except Exception as e:
            ok, why = False, f"{type(e).__name__}: {e}"
        if not all_agree(ok):
            log(f"rq_query_batch_sharded_device failed in warm-up ({why or 'on another rank'}): falling back to torch.distributed")
            if keep_alive and isinstance(keep_alive[0], sharding.HostCollectives):
                sharding.HostCollectives.uninstall()
            collective_path = "per-call entries + torch.distributed (" + args.backend + "), after the C-ABI step failed: " + (why or "another rank")
            pc_local = torch.zeros((B, nprobe), device=dev, dtype=torch.int32)
            pd_local = torch.zeros((B, nprobe), device=dev, dtype=torch.float32)
            seeded = sharding.SeededShardQuery(B, topk, dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    prof = {}

    def run_steps(count, record, depth=depth, rotate=False):
        """`count` steps; with depth > 1 a step's batch is begun before the previous one is ended, so consecutive
        batches overlap on the device (each owns a workspace, a stream and an output slot).  rotate: step i answers query set i mod 4
        (otherwise set 0)."""
        res = None
        if depth == 1:
            for it in range(count):
                qsel[0] = it % len(query_sets) if rotate else 0
                extra_prof.clear()
                ts = time.perf_counter()
                res = step()
                if os.environ.get("RQ_BENCH_STEP_TIMES"):   # diagnostics: wall time of every step (the engine call is synchronous)
                    log(f"step {'timed' if record else 'warm-up'}: {(time.perf_counter() - ts) * 1e3:.1f} ms, "
                        f"retries so far {rqi.last_profile()['retries']}")
                if record:
                    for pr in [rqi.last_profile()] + extra_prof:
                        for key, v in pr.items():
                            prof[key] = prof.get(key, 0) + v
                        prof["survivor_workspace_bytes_max"] = max(prof.get("survivor_workspace_bytes_max", 0), pr.get("survivor_workspace_bytes", 0))
            return res
        pending = []
        for i in range(count + depth - 1):
            if i < count:
                od, oi, on = outs[i % depth]
                pending.append((idx.query_batch_device_begin(query_sets[i % len(query_sets) if rotate else 0].data_ptr(), B, d, nprobe, topk, od.data_ptr(),
                                                             oi.data_ptr(), on.data_ptr()), i % depth))
            if i >= depth - 1:
                tk, slot = pending.pop(0)
                idx.query_batch_device_end(tk)
                if record:
                    for key, v in rqi.last_profile().items():
                        prof[key] = prof.get(key, 0) + v
                od, oi, on = outs[slot]
                res = (od, oi.to(torch.int64) & 0xFFFFFFFF, on)
        return res

    rqi.set_profiling(2)         # HIP events on the engine's own stream around the scan launches (the roofline kernel)
    run_steps(args.warmup, False)
    fence()
    rabitq_amd.metrics_reset()
    t1 = time.perf_counter()
    run_steps(args.steps, True, rotate=True)
    fence()
    elapsed = time.perf_counter() - t1
    qsel[0] = 0
    # the same loop with two batches in flight: one batch's HBM-bound stages overlap the other's compute-bound scan
    overlap = None
    if not sharded and args.two_in_flight and extras:
        run_steps(2, False, 2)
        fence()
        t2 = time.perf_counter()
        run_steps(args.steps, False, 2, rotate=True)
        fence()
        e2 = time.perf_counter() - t2
        overlap = {"batches_in_flight": 2, "value": round(B * args.steps / e2, 1), "unit": "queries/s",
                   "ms_per_step": round(e2 / args.steps * 1e3, 3),
                   "note": "rq_query_batch_device_begin/_end; per-kernel times are not clean in this mode, so the headline "
                           "`value` and the roofline figures are taken with one batch at a time"}
    # per-kernel-group breakdown from ONE extra, untimed step (an event pair per group costs ~10 us of stream time each)
    rqi.set_profiling(1)
    extra_prof.clear()
    qsel[0] = 0
    res = step()   # (set 0: also the batch whose first `ngt` queries have a ground truth)
    fence()
    breakdown = {}
    for pr in [rqi.last_profile()] + extra_prof:   # a multi-GPU step is two engine calls
        for key, v in pr.items():
            if key.startswith("ms_"):
                breakdown[key[3:]] = round(breakdown.get(key[3:], 0.0) + v, 3)
    # share of the matrix-core scan's 32x32 sub-tile steps that were flagged by the integer gate and took the exact f32 path
    # (always-on counters of the timed steps)
    ep = {"matrix_exact_steps": prof.get("matrix_exact_steps", 0), "matrix_subtile_steps": prof.get("matrix_subtile_steps", 0)}
    exact_rate = ep["matrix_exact_steps"] / ep["matrix_subtile_steps"] if ep["matrix_subtile_steps"] else None
    rqi.set_profiling(2)
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if args.backend == "gloo" else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    qps = B * args.steps / elapsed

    # ---- recall@topk on the ground-truth subset ------------------------------------------------------
    rd, ri, rn = res
    ri = ri.cpu().numpy()
    recall = float(np.mean([len(set(ri[q, :topk].tolist()) & set(gt[q].tolist())) / topk for q in range(ngt)]))
    m = rabitq_amd.metrics()

    # ---- roofline of the dominant kernel ----------------------------------------------------------------
    # The scan is two kernels.  With 10 000 queries per batch every list is shared by ~150 queries, so the
    # launch that covers ~97 % of the candidates (scan_mfma_kernel) is bound by the matrix/vector pipes, not by
    # HBM: it is priced in flops against the fp6 MFMA peak, with its algorithmic byte rate and the PMC-measured
    # HBM traffic next to it.  The HBM-bound regime of the scan (a small batch, no sharing) is measured below.
    scan_s = prof["ms_scan"] * 1e-3
    launches = max(prof["scan_launches"], 1)
    achieved = prof["scan_bytes"] / scan_s / 1e9 if scan_s > 0 else 0.0
    # HBM bytes are NOT measured in this run (PMC counters need their own rocprofv3 pass): when the committed PMC summary
    # was taken on this exact workload its figures are quoted, labelled as such; otherwise traffic is null.
    traffic, dom_traffic, traffic_src = None, None, None
    import glob as _glob
    want_cfg = {"vectors": n, "dim": d, "lists": k_local, "nprobe": nprobe, "batch": B, "distribution": "hard" if hard else "easy"}
    for tr_path in sorted(_glob.glob(os.path.join(ROOT, "profiles", "scan_traffic*.json"))):
        try:
            tj = json.load(open(tr_path))
            cfg_t = dict(tj.get("config", {}))
            cfg_t.setdefault("distribution", "easy")
            if cfg_t == want_cfg:  # a PMC pass of THIS workload only
                traffic = tj.get("hbm_bytes_per_launch")
                dom_traffic = tj.get("dominant_launch", {}).get("hbm_read_bytes")
                traffic_src = (f"committed PMC pass profiles/{os.path.basename(tr_path)} (FETCH_SIZE x2, same workload; its dominant launch: "
                               f"{tj.get('dominant_launch', {}).get('kernel', '?')}), not measured in this run")
                break
        except Exception:
            continue
    scan_all = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "all scan launches of a batch: scan_kernel<W,CPL> (early stages) + scan_mfma_kernel<W,NT>",
                "launches": int(launches), "avg_launch_ms": round(prof["ms_scan"] / launches, 4),
                "algorithmic_bytes_per_launch": int(prof["scan_bytes"] / launches),
                "note": "achieved = algorithmic bytes (sum over probed lists of len*(dim/8+16) per query) / scan "
                        "kernel time from HIP events; a list is read from HBM once per launch and scored against "
                        "every query probing it, so the algorithmic rate exceeds physical HBM traffic"}
    PEAK_FP6 = 256 * 4 * 2.4e9 * (2 * 32 * 32 * 64 / 32) / 1e12   # v_mfma_f32_32x32x64_f8f6f4 (fp6): 32 cycles per SIMD
    if prof.get("matrix_launches", 0) > 0 and prof["ms_scan_matrix"] > 0:
        ml, mm, mp = prof["matrix_launches"], prof["ms_scan_matrix"] * 1e-3, prof["matrix_pairs"]
        flops = mp * 2.0 * idx.dim                       # USEFUL work only: the dim-long integer dot product of every pair
        additive = prof.get("matrix_additive_launches", 0) >= ml > 0   # every matrix-core launch ran the additive gate
        over = 0.0 if additive else mp * 2.0 * 16        # overhead: the 16-slot bf16 threshold MFMA per pair (not counted)
        mbytes = mp * (idx.dim / 8 + 16)
        roofline = {"bound": "mfma", "achieved": round(flops / mm / 1e12, 1), "peak": round(PEAK_FP6, 1), "unit": "TFLOP/s",
                    "frac": round(flops / mm / 1e12 / PEAK_FP6, 4), "traffic": dom_traffic, "traffic_source": traffic_src,
                    # fabric bytes of the launch (FETCH_SIZE x2: incl. Infinity-Cache hits) over the bytes of the index it reads once
                    # (codes + factors of every vector): what the per-XCD L2s fetch more than once, plus the query tile images
                    "traffic_over_unique_index_bytes": None if not dom_traffic else round(dom_traffic / (idx.n * (idx.dim / 8 + 16)), 2),
                    "kernel": ("scan_mfma_kernel<W,NT,ADD> (v_mfma_f32_32x32x64_f8f6f4 e2m3 x e2m1; additive gate S* >= B_q + G_c, no threshold MFMA)"
                               if additive else "scan_mfma_kernel<W,NT> (v_mfma_f32_32x32x64_f8f6f4 e2m3 x e2m1 + bf16 threshold MFMA)"),
                    "gate": "additive" if additive else "bf16 rank-5 threshold",
                    "launches": int(ml), "avg_launch_ms": round(mm / ml * 1e3, 4),
                    "algorithmic_flops_per_launch": int(flops / ml), "pairs_per_launch": int(mp / ml),
                    "overhead_flops_per_launch_threshold_mfma": int(over / ml),
                    "algorithmic_bytes_per_launch": int(mbytes / ml),
                    "algorithmic_GBps": round(mbytes / mm / 1e9, 1), "algorithmic_GBps_over_hbm_peak": round(mbytes / mm / 1e9 / 8000.0, 2),
                    "note": "achieved = (query, candidate) pairs scored x 2*dim useful flops / launch time (HIP events); the "
                            "threshold MFMA is overhead and listed separately; the launch is shared-list compute: its algorithmic "
                            "byte rate is far above the 8 TB/s HBM peak and the PMC-measured HBM traffic far below the algorithmic bytes"}
    else:
        roofline = scan_all

    # ---- the same scan kernel in its HBM-bound regime: a small batch, (almost) no list shared --------
    small = []
    for sb in [min(int(v), B) for v in str(args.small_batch).split(",") if extras and v.strip() and int(v) > 0]:
        small.append(small_batch_regime(idx, queries, sb, d, nprobe, topk, out_d, out_i, out_n, n, k_local))
    small = small or None

    # ---- one query at a time through the host-pointer API, as crates/cli/src/main.rs:69-75 does -------
    single = None
    if rank == 0 and extras:
        qh = queries[:64].cpu().numpy()
        for q1 in qh[:4]:
            idx.query(q1, nprobe, topk)
        lat = []
        for q1 in qh:
            t2 = time.perf_counter()
            idx.query(q1, nprobe, topk)
            lat.append(time.perf_counter() - t2)
        lat = np.array(lat)
        single = {"queries": len(lat), "mean_ms": round(float(lat.mean() * 1e3), 4),
                  "p50_ms": round(float(np.median(lat) * 1e3), 4), "p99_ms": round(float(np.quantile(lat, 0.99) * 1e3), 4),
                  "queries_per_s": round(len(lat) / float(lat.sum()), 1), "device_ms": round(rqi.last_profile()["ms_total"], 4)}

    # ---- the drop-in signature: RaBitQ::query takes &[f32] from HOST memory (src/rabitq.rs:268-274).  rq_query_batch with host
    # buffers = the timed step + PCIe both ways (B*len*4 bytes in, B*topk*8 + B*4 out), pageable and pinned ---------------------
    host_entry = None
    if rank == 0 and extras and not sharded:
        from rabitq_amd import _lib as _l
        import ctypes as C
        host_entry = {}
        for kind in ("pageable", "pinned"):
            pin = kind == "pinned"
            hq = torch.empty((B, d), dtype=torch.float32, pin_memory=pin)
            hq.copy_(queries)
            hd = torch.empty((B, topk), dtype=torch.float32, pin_memory=pin)
            hi = torch.empty((B, topk), dtype=torch.int32, pin_memory=pin)
            hn = torch.empty((B,), dtype=torch.int32, pin_memory=pin)
            call = lambda: _l.check(_l.lib().rq_query_batch(idx._h, C.c_void_p(hq.data_ptr()), B, d, nprobe, topk, 0,   # noqa: E731
                                                            C.c_void_p(hd.data_ptr()), C.c_void_p(hi.data_ptr()), C.c_void_p(hn.data_ptr())))
            call()
            hs = min(args.steps, 5)
            th = time.perf_counter()
            for _ in range(hs):
                call()
            eh = (time.perf_counter() - th) / hs
            same = bool(torch.equal(hi.to(torch.int64) & 0xFFFFFFFF, res[1].cpu()))
            host_entry[kind] = {"ms_per_step": round(eh * 1e3, 3), "queries_per_s": round(B / eh, 1), "steps": hs,
                                "ids_equal_to_device_entry": same}
            del hq, hd, hi, hn
        host_entry["entry"] = "rq_query_batch (host pointers in and out; one blocking call per step)"
        host_entry["bytes_in_out_per_step"] = [B * d * 4, B * topk * 8 + B * 4]

    # ---- between the small-batch regimes and the headline batch: what a service's request sizes would see -------------------
    sweep = None
    default_sweep_beyond = (n, d, args.distribution) == (100_000_000, 128, "easy") and B == 65536   # (also two sizes beyond the headline batch)
    if rank == 0 and extras and not sharded and args.batch_sweep:
        sweep = batch_sweep(idx, centres, qweights, args.sigma, dev, d, nprobe, topk,
                            [b for b in (128, 512, 2048, 8192, 16384) if b < B] + ([2 * B, 4 * B] if default_sweep_beyond else []))

    size_txt = f"{n // 1_000_000}Mx{d}" if n % 1_000_000 == 0 else f"{n}x{d}"
    line = {"metric": f"queries/sec at recall@10>=0.95, {size_txt}; HBM GB/s on popcount scan", "value": round(qps, 1),
            "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak",   # per-GPU work is fixed: every rank indexes --vectors rows and 4096 lists of its own, and the batch grows with N
            "scaling_note": f"weak: {n // 1_000_000}M vectors and {k_local} lists per GPU at every N; one step answers {B} queries "
                            f"(65536 x N by default) against ALL {k} lists, so a rank scans as many (query, list) pairs per step as "
                            "one GPU does alone; the coarse ranking is sliced by QUERIES over the ranks (rank r ranks queries "
                            "[r B/N, (r+1) B/N) against ALL lists: (B/N) x k distances per rank, i.e. per-rank coarse work grows with N "
                            "through k only; one all-gather carries B/N x nprobe (distance, list) keys per rank)",
            "vs_baseline": None, "dtype": "exact integer dot (fp6 MFMA, v_dot8_u32_u4) + f32", "data": "synthetic",
            "engine_options": args.option, "collective_path": collective_path,
            "config": {"workload": f"{n // 1_000_000}Mx{d} synthetic mixture per GPU, {k_local} lists per GPU, "
                                   f"nprobe={nprobe}, topk={topk}, batch={B}" +
                                   (" (BASELINE.json configs[2])" if (n, d, k_local, nprobe, world, args.distribution) == (100_000_000, 128, 4096, 64, 1, "easy") else "") +
                                   (" (BASELINE.json configs[3])" if (n, d, k_local, nprobe, world, args.distribution) == (100_000_000, 768, 4096, 64, 1, "easy") else "") +
                                   (f" = {world * n // 1_000_000}M x {d} over {k} lists on {world} GPUs" if world > 1 else "") +
                                   (f" -- REHEARSAL of one rank of a {args.emulate_world}-GPU run ({k} lists in all, this rank owns {k_local}; "
                                    f"coarse ranking unsliced, recall against this rank's own vectors only)" if getattr(args, "emulate_world", 1) > 1 else "") +
                                   (" (BASELINE.json configs[4]: 1B x 128; k is unspecified there: 32 768 = 4096 per GPU, SURVEY.md 8d)"
                                    if (n, d, k_local, nprobe, world, args.distribution) == (125_000_000, 128, 4096, 64, 8, "easy") else "") +
                                   (f"; batch = 65536 x {world} GPUs" if world > 1 and B == 65536 * world else ""),
                       "batches_in_flight": depth,
                       "n_per_gpu": n, "dim": d, "lists_total": k, "nprobe": nprobe, "topk": topk, "batch": B,
                       "sigma": args.sigma, "centre_scale": centre_scale, "sharding": f"vectors x{world}",
                       "distribution": args.distribution if not hard else
                       f"hard: centre std = {args.hard_centre_ratio} x sigma (overlapping clusters), Zipf({args.zipf}) list sizes, "
                       f"centroids trained by rq_kmeans_device"},
            "recall_at_10": round(recall, 4), "recall_queries": ngt, "build_seconds": round(build_s, 2), "build": build_info,
            "rough_per_query": m["rough"] / max(m["query"], 1), "precise_per_query": m["precise"] / max(m["query"], 1),
            "kernel_ms_per_step": breakdown, "scan_ms_per_step_timed": round(prof["ms_scan"] / args.steps, 3),
            "rerank_candidates_per_query": prof["rerank_candidates"] / (B * args.steps),
            # of those: rejected on the fp16 shadow row alone (2*dim bytes read instead of 4*dim; results identical)
            "rerank_shadow_rejects_per_query": prof["rerank_shadow_rejects"] / (B * args.steps),
            "matrix_exact_path_rate": None if exact_rate is None else round(exact_rate, 5),
            "retries": int(prof["retries"]),
            "query_batches": {"distinct_batches_in_timed_steps": len(query_sets),
                              "note": "warm-up on batch 0 only; timed step i answers batch i mod 4 (fresh draws from the same mixture); "
                                      "`retries` = queries re-run in the timed steps because a survivor capacity learnt on batch 0 did not hold"},
            "survivor_workspace_GB": round(prof.get("survivor_workspace_bytes_max", 0) / 1e9, 2),
            "segmented_passes_per_step": prof.get("segmented_passes", 0) / args.steps,
            "roofline": roofline, "roofline_scan_all_launches": scan_all,
            "roofline_rotation": rotation,
            "scan_small_batch": small, "batch_sweep": sweep, "single_query": single, "two_batches_in_flight": overlap, "host_entry": host_entry}

    # ---- CPU baseline: the oracle (port of the reference's AVX2 path) on this box's host cores ------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_queries > 0 and extras:
        if idx.n * idx.dim * 4 > (150 << 30):
            # the CPU path needs all raw vectors in host memory (src/rabitq.rs:59); beyond what the box allows a process
            line["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 1, "kind": "port",
                                    "sample": f"skipped: {idx.n * idx.dim * 4 / 1e9:.0f} GB of raw vectors do not fit the host memory "
                                              "a process may use here; measured on the 100Mx128 configuration instead"}
        else:
            line["cpu_baseline"] = cpu_baseline(idx, queries[:args.cpu_queries].cpu().numpy(), nprobe, topk, ri, d)
    # everything this workload holds on the device goes before the next one is built
    if keep_alive and isinstance(keep_alive[0], sharding.HostCollectives) and sharded:
        sharding.HostCollectives.uninstall()
    for obj in keep_alive:
        if hasattr(obj, "close"):
            obj.close()
    idx.close()
    del outs, out_d, out_i, out_n, queries, query_sets, res, rd, rn
    torch.cuda.empty_cache()
    return line


def batch_sweep(idx, centres, qweights, sigma, dev, d, nprobe, topk, sizes):
    """Queries per call between the small-batch path and the headline batch (crates/service/src/main.rs:36-44 answers whatever a
    request brings): every call answers queries the engine has not seen (six distinct draws per size, one of them as warm-up)."""
    import torch
    from rabitq_amd import index as rqi
    from tests import synth
    rows = []
    for b in sizes:
        sets = [synth.device_queries(centres, b, sigma, dev, seed=50_000 + 17 * b + i, weights=qweights) for i in range(6)]
        od = torch.empty((b, topk), device=dev, dtype=torch.float32)
        oi = torch.zeros((b, topk), device=dev, dtype=torch.int32)
        on = torch.zeros((b,), device=dev, dtype=torch.int32)
        idx.query_batch_device(sets[0].data_ptr(), b, d, nprobe, topk, od.data_ptr(), oi.data_ptr(), on.data_ptr())
        torch.cuda.synchronize()
        retries, small_passes, matrix = 0, 0, 0
        per_call = []
        for q in sets[1:]:
            t0 = time.perf_counter()
            idx.query_batch_device(q.data_ptr(), b, d, nprobe, topk, od.data_ptr(), oi.data_ptr(), on.data_ptr())   # (returns with the results in place)
            per_call.append(time.perf_counter() - t0)
            pr = rqi.last_profile()
            retries += pr["retries"]
            small_passes += pr["small_batch_passes"]
            matrix += pr["matrix_launches"]
        torch.cuda.synchronize()
        el = sorted(per_call)[len(per_call) // 2]   # the median call: a call that happens to grow a workspace buffer is listed, not averaged in
        rows.append({"batch": b, "ms_per_call": round(el * 1e3, 3), "ms_per_call_max": round(max(per_call) * 1e3, 3), "queries_per_s": round(b / el, 1), "retries": int(retries),
                     "path": "small-batch kernels" if small_passes else ("staged: VALU early stages + matrix-core final stage" if matrix else "staged: VALU scan only")})
        del sets, od, oi, on
    for a, c in zip(rows, rows[1:]):
        c["queries_per_s_vs_previous_size"] = round(c["queries_per_s"] / a["queries_per_s"], 2)
    return rows


def small_batch_regime(idx, queries, sb, d, nprobe, topk, out_d, out_i, out_n, n, k_local):
    """The scan kernel in its HBM-bound regime: a small batch, (almost) no list shared between queries."""
    from rabitq_amd import index as rqi
    nqs = queries.shape[0]
    reps = 10
    for w in range(2):   # warm-up on queries the timed calls do not use (the tail of the batch), so no timed call finds its lists cached
        qw = queries[max(0, nqs - (w + 1) * sb): max(sb, nqs - w * sb)]
        idx.query_batch_device(qw.data_ptr(), sb, d, nprobe, topk, out_d.data_ptr(), out_i.data_ptr(), out_n.data_ptr())
    sp = {}
    for r in range(reps):
        q0 = (r * sb) % max(1, nqs - 3 * sb + 1)      # different queries per call, all before the warm-up slice
        idx.query_batch_device(queries[q0:q0 + sb].data_ptr(), sb, d, nprobe, topk, out_d.data_ptr(), out_i.data_ptr(),
                               out_n.data_ptr())
        for key, v in rqi.last_profile().items():
            sp[key] = sp.get(key, 0) + v
    gbs = sp["scan_bytes"] / (sp["ms_scan"] * 1e-3) / 1e9
    small = {"batch": sb, "scan_ms_per_batch": round(sp["ms_scan"] / reps, 4),
             "total_ms_per_batch": round(sp["ms_total"] / reps, 4),
             "scan_algorithmic_GBps": round(gbs, 1), "algorithmic_frac_of_8TBps": round(gbs / 8000.0, 4),
             # the whole call (every launch, HIP events around the pass): algorithmic bytes over its device time
             "whole_call_algorithmic_GBps": round(sp["scan_bytes"] / (sp["ms_total"] * 1e-3) / 1e9, 1),
             "whole_call_frac_of_8TBps": round(sp["scan_bytes"] / (sp["ms_total"] * 1e-3) / 1e9 / 8000.0, 4),
             "path": "few-launch small-batch path (kernels_small.h)" if sp.get("small_batch_passes", 0) else "staged path",
             "timing": "HIP events of the engine around its scan launches / the whole pass, fresh queries in every call (an event "
                       "pair adds ~4 us to a launch: at batch 1 the 50 MB scan launch itself is ~7-10 us, see the kernel-trace "
                       "figures under committed_profile)",
             "queries_per_s": round(sb * reps / (sp["ms_total"] * 1e-3), 1)}
    # physical HBM rate of the same regime: PMC FETCH_SIZE (x2) over kernel-trace durations, committed profile
    # (the newest committed pass; its round is part of `source`: a pass taken on an earlier round's kernels must say so)
    hp = next((os.path.join(ROOT, "profiles", f) for f in ("r05_hbm_regime.json", "r04_hbm_regime.json", "r03_hbm_regime.json", "r02_hbm_regime.json")
               if os.path.exists(os.path.join(ROOT, "profiles", f))), "")
    if hp:
        try:
            for wl in json.load(open(hp))["workloads"]:
                c = wl["config"]
                if (c["vectors"], c["dim"], c["lists"], c["nprobe"]) == (n, d, k_local, nprobe):
                    for rg in wl["regimes"]:
                        if rg["batch"] == sb:   # NOT measured in this run: kept apart from the in-run figures above
                            small["committed_profile"] = {
                                "source": os.path.basename(hp) + " (rocprofv3 --pmc FETCH_SIZE x2 / kernel-trace time)",
                                "source_round": os.path.basename(hp)[:3],
                                "fabric_GBps_all_scan_launches": round(rg["physical_GBps"], 1),
                                "fabric_frac_of_8TBps_all_scan_launches": round(rg["physical_frac_of_8TBps"], 4),
                                "label": "fabric bytes: FETCH_SIZE includes Infinity-Cache hits, so this can exceed what HBM itself "
                                         "delivers (~6.3 TB/s); x2 applied because the scan's loads are 16 B / lane, coalesced"}
        except Exception:
            pass
    return small


def self_launch(n: int) -> int:
    """One child process per rank, the environment torch.distributed.run would have set; rank 0's stdout (the JSON
    line) and everybody's stderr pass through.  Equivalent to
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # Watchdog: a rank stuck in communicator set-up (ncclCommInitRank waits for every rank) or behind a rank that died must not hold
    # the run until somebody else's limit.  On expiry -- or as soon as one rank has failed -- the children are terminated (then
    # killed) and the parent exits non-zero; nothing is ever re-exec'd from a process that touched the GPU.
    deadline = time.time() + float(os.environ.get("RQ_BENCH_DEADLINE_S", "1500"))
    rc, live = 0, list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is not None:
                live.remove(pr)
                rc = max(rc, abs(code))
        if not live:
            break
        if rc != 0 or time.time() > deadline:
            why = "a rank failed" if rc != 0 else "deadline (RQ_BENCH_DEADLINE_S) passed"
            print(f"[bench] {why}: terminating {len(live)} rank process(es)", file=sys.stderr, flush=True)
            for pr in live:
                pr.terminate()
            t_end = time.time() + 20
            for pr in live:
                try:
                    pr.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    pr.kill()
                    pr.wait()
            return rc or 124
        time.sleep(0.2)
    return rc


def cpu_baseline(idx, qs, nprobe, topk, gpu_ids, d):
    """Times oracle.rqo_query (single thread, sequential queries: exactly how crates/cli/src/main.rs:69-80
    measures QPS) on a bounded sample, on an index the GPU engine built and handed over."""
    import oracle
    t0 = time.time()
    arrays = dict(base=idx.base, orthogonal=idx.orthogonal, centroids=idx.centroids, offsets=idx.offsets,
                  map_ids=idx.map_ids, codes=idx.codes, factors=idx.factors)
    log(f"index copied to host for the CPU baseline ({time.time() - t0:.1f}s)")
    ov = oracle.OracleIndex.view(idx.dim, **arrays)
    oracle.metrics_reset()
    agree = 0
    per_q = []
    budget_s = 25.0
    tstart = time.perf_counter()
    done = 0
    for qi, q in enumerate(qs):
        t1 = time.perf_counter()
        od, oi = ov.query(q, nprobe, topk)
        per_q.append(time.perf_counter() - t1)
        agree += int(set(oi.tolist()) == set(int(v) for v in gpu_ids[qi, :topk]))
        done += 1
        if time.perf_counter() - tstart > budget_s:
            break
    m = oracle.metrics()
    qps1 = done / sum(per_q)
    # service-style: all host cores, one query per thread (crates/service/src/main.rs:36-44)
    cores = min(os.cpu_count() or 1, 64)
    scratch = [np.empty(idx.max_list_len + 1, np.float32) for _ in range(cores)]
    reps = max(1, min(4, int(8.0 / max(sum(per_q) / cores, 1e-3))))
    work = [qs[i % done] for i in range(done * reps)]

    def run(tid):
        for q in work[tid::cores]:
            ov.query(q, nprobe, topk)
    t2 = time.perf_counter()
    th = [threading.Thread(target=run, args=(t,)) for t in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    qps_all = len(work) / (time.perf_counter() - t2)
    # scan-only rate (calculate_rough_distance loop alone) for the algorithmic GB/s of the CPU path
    t3 = time.perf_counter()
    scanned = sum(ov.scan_only(q, nprobe, scratch[0]) for q in qs[:min(done, 8)])
    scan_gbs = scanned * (d // 8 + 16) / (time.perf_counter() - t3) / 1e9
    return {"value": round(qps1, 2), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"first {done} queries of the batch, full query() one at a time on 1 thread "
                      f"(crates/cli/src/main.rs:69-80 method), index built by the GPU engine",
            "all_cores_value": round(qps_all, 1), "all_cores": cores,
            "scan_only_algorithmic_GBps_1thread": round(scan_gbs, 2),
            "ids_identical_to_gpu": f"{agree}/{done}",
            "rough_per_query": m["rough"] / max(m["query"], 1), "precise_per_query": m["precise"] / max(m["query"], 1)}


if __name__ == "__main__":
    main()
