#!/usr/bin/env python3
"""bench.py — decode tok/s + %HBM-roofline, Llama-3.2-1B Q8_0 greedy decode on MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one decode step (one token through the whole hot path: embedding row, 16 decoder layers,
classifier, argmax).  Workload (config.workload): BASELINE.json configs[1] — synthetic LMRS image with the
real Llama-3.2-1B shapes (tools/synth_lmrs.py, seed 1234, weights quantised with the reference's Q8_0
quantiser), a 16-token synthetic prompt (= the W warm-up steps by default) and then K greedy tokens.
Weights, KV cache and every activation are resident in HBM when the timed region starts; token ids never
leave the device between steps.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the dominant kernel family (the fused Q8_0 dequant-GEMV, lmrs::gemv_static_kernel<...>): algorithmic
                bytes (int8 weights + f32 group scales) of one decode step's 65 GEMV launches / their summed
                durations AS THEY RUN INSIDE THE STEP: the real step (same launches as the captured graph, live
                activations and positions) is replayed eagerly with a HIP event pair on every dispatch
                (lmrs_bench_step), against the 8 TB/s HBM3E peak; `frac_vs_measured_copy` prices the same figure
                against the 6.29 TB/s the chip sustains on a device copy.  `in_step` lists every kernel of the step
                (attention and argmax included).  `achieved` / `frac` are the WHOLE-STEP figure (SURVEY.md §8d bytes
                per token / measured time per token, = `path`); the family's own figure on its event durations,
                which leaves the gaps between launches out, is `family_achieved` / `family_frac`.  `traffic` comes from a committed rocprofv3 PMC summary and is null
                unless that summary was taken for this model, quantisation and build of the kernels.
  cpu_baseline  the CPU oracle (a C port of the reference's arithmetic, oracle/lmrs_oracle.c; the Rust
                reference itself cannot be built here) timed on this box's host cores on the same prompt.
                The same run is the parity gate: the K token ids must be identical.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
MEASURED_COPY_GBPS = 6290.0     # what the chip sustains on a device copy (same guide: "~6.3 TB/s achievable")


def exchange_unique_id(dist, rank, make_id):
    """Rank 0 makes the 128-byte communicator id, every rank receives it (any torch.distributed backend)."""
    box = [make_id() if rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    uid = box[0]
    assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
    return bytes(uid)


def kernel_source_hash():
    """Identifies the build of the kernels: sha256 over the sources liblmrs_hip.so is made from and the Makefile that carries its flags."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lm.rs_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc", ".h", ".cpp")) or f == "Makefile":
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(model, qtype):
    """HBM bytes per GEMV launch from a committed rocprofv3 PMC summary of this same command (profiles/*_traffic*.json, made by
    tools/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md applied).  bench.py cannot
    collect hardware counters itself: the figure is reported only when a summary exists for THIS model, quantisation and build of
    the kernels (source hash); otherwise null."""
    want = (model, qtype, kernel_source_hash())
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if "traffic" not in f or not f.endswith(".json"):
            continue
        try:
            doc = json.load(open(os.path.join(pdir, f)))
            if (doc.get("model"), doc.get("qtype"), doc.get("kernel_source_hash")) != want:
                continue
            k = {n: v for n, v in doc["kernels"].items() if "gemv" in n or "qkv_attn" in n}     # the dequant-GEMV family (the qkv launch carries the attention workgroups)
            n = sum(v["launches"] for v in k.values())
            return round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in k.values()) / n) if n else None
        except (OSError, ValueError, KeyError):
            continue
    return None


def rocprof_family_us(model, qtype):
    """Time of the dequant-GEMV family per decode step priced with rocprofv3's per-kernel averages (profiles/*_rocprof_*.json, made by
    tools/rocprof_summary.py from `rocprofv3 --kernel-trace --stats -- python bench.py --cpu-steps 0`), for THIS model, quantisation
    and build of the kernels; None otherwise."""
    want = (model, qtype, kernel_source_hash())
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if "_rocprof_" not in f or not f.endswith(".json"):
            continue
        try:
            doc = json.load(open(os.path.join(pdir, f)))
            if (doc.get("model"), doc.get("qtype"), doc.get("kernel_source_hash")) == want:
                return float(doc["gemv_family_us_per_step"])
        except (OSError, ValueError, KeyError):
            continue
    return None


def shard_description(lmrs_amd, model, world, transport):
    """Which matrices the library split for this model and world size (lmrs_shard_plan: it row-splits the layers' matrices only when
    the stream a shard stops reading outweighs two exchanges per layer; below that every GPU runs the layers whole and only the
    classifier's rows are split - DESIGN.md section 6)."""
    plan = lmrs_amd.shard_plan(model.args, 0, world)
    if plan["q_heads"][1] == model.args.n_heads:
        return (f"cls{world}: every GPU runs the layers whole (no exchange inside a layer), the classifier's rows are split over {world} GPUs, "
                f"one exchange of argmax partials per token by {transport}")
    return f"tp{world}: rows of every weight matrix split over {world} GPUs, slices exchanged by {transport}"


def max_over_ranks(dist, seconds, device=None):
    """The contract's timing rule: the step time of the job is the slowest rank's."""
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def headline_then_guarded(headline, second, barrier, rank, limit_s, leave=os._exit):
    """N > 1: run the headline configuration, hold its line, then run the further configuration(s) under a watchdog on every rank; rank 0
    prints exactly ONE JSON line whatever they do.  `second`: a callable (its figures ride along as `library_choice`) or a list of
    (key, callable) run in order, each with its own limit_s.  A run returning: the line carries its figures under its key.  A run raising
    on this rank, or not returning within limit_s: the line carries `<key>: {value: null, skipped: why}` (so do the runs behind it) and the
    process leaves through `leave` (os._exit: the abandoned run's collectives may never return, so no orderly teardown)."""
    import threading
    out = headline()
    barrier()
    extras = [("library_choice", second)] if callable(second) else list(second)
    printed = threading.Lock()
    state = {"key": extras[0][0] if extras else None}

    def emit_and_leave(why):
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            hit = False
            for key, _ in extras:                               # the run that failed and everything behind it
                hit = hit or key == state["key"]
                if hit:
                    out[key] = {"value": None, "skipped": why if key == state["key"] else f"not run: `{state['key']}` did not finish"}
            print(json.dumps(out), flush=True)
        sys.stdout.flush(); sys.stderr.flush()
        leave(0)

    for key, run in extras:
        state["key"] = key
        dog = threading.Timer(limit_s, emit_and_leave, args=(f"the {key.replace('_', '-')} run did not finish within {limit_s:.0f} s",))
        dog.daemon = True; dog.start()
        try:
            ex_out = run()
            barrier()
            figures = None
            if rank == 0 and ex_out is not None:                # (inside the guard: a run that returns without these keys must not cost the headline line)
                figures = {k: ex_out.get(k) for k in ("value", "ms_per_step", "transport", "rccl_nranks", "parity")}
                figures["parallelism"] = ex_out["config"]["parallelism"]
                figures["roofline"] = {k: ex_out["roofline"].get(k) for k in ("frac", "redundant_bytes_per_step", "bytes_streamed_per_gpu_per_step", "step_split")}
        except BaseException as e:                              # noqa: BLE001 - the headline line must get out whatever happened here
            print(f"[rank {rank}] {key.replace('_', '-')} run failed: {type(e).__name__}: {e}", file=sys.stderr)
            emit_and_leave(f"the {key.replace('_', '-')} run failed on a rank: {type(e).__name__}")
            time.sleep(limit_s)                                 # (the watchdog thread is printing: wait for its exit)
            return out
        dog.cancel()
        if figures is not None:
            out[key] = figures
    if printed.acquire(blocking=False) and rank == 0:
        print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="llama-3.2-1b")
    ap.add_argument("--qtype", default="q8_0", choices=["q8_0", "q4_0"], help="weight quantisation of the synthetic image (config 3: gemma-2-2b q4_0)")
    ap.add_argument("--cpu-steps", type=int, default=-1, help="oracle steps for cpu_baseline (-1: same as the GPU run, 0: skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--vision", action="store_true", help="BASELINE configs[4] (use with --model phi-3.5): add a `vision` object - CLIP tower, projector, fill_kv_cache(320)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # LMRS_BENCH_FORCE_DIST=1: take the N > 1 code path (torch.distributed + RCCL communicator + sharded context) with a world of
    # one - the only way to exercise that path, torch's bundled HIP / RCCL runtimes included, on a single-GPU box
    force_dist = os.environ.get("LMRS_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        # LMRS_BENCH_ONE_DEVICE=1 (verification on a one-GPU box): every rank uses device 0 and the process group runs on gloo
        # (RCCL refuses two ranks on one device); the shards then exchange through the peer-to-peer transport, as on a real node.
        one_device = os.environ.get("LMRS_BENCH_ONE_DEVICE") == "1"
        if one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    hip = ctypes.CDLL("libamdhip64.so")

    def device_sync():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
        else:
            hip.hipDeviceSynchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def all_ranks_ok(ok):
        import torch
        flag = torch.tensor([1 if ok else 0], device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    import lmrs_amd
    from tools import synth_lmrs as S
    lmrs_amd.build()                       # (re)compile liblmrs_hip.so if it is missing or older than its sources (a fresh clone has none)

    cfg = S.CONFIGS[args.model]
    K, W = args.steps, max(1, args.warmup)
    if W + K > 8192:
        raise SystemExit("warmup + steps must fit the 8192-position KV cache")
    t0 = time.time()
    qt = S.Q8_0 if args.qtype == "q8_0" else S.Q4_0
    # LMRS_BENCH_IMAGE_CACHE=dir: keep / reuse the synthetic image as a file (the profiling recipe builds it once outside rocprofv3, whose
    # tool library segfaults under the image builder's thread pool for the larger models; same bytes either way: seeded)
    cache = os.environ.get("LMRS_BENCH_IMAGE_CACHE")
    cpath = os.path.join(cache, f"{cfg.name}_{args.qtype}_seed1234.lmrs") if cache else None
    if cpath and os.path.exists(cpath) and os.path.getsize(cpath) == S.image_size(cfg, qt):
        img = np.fromfile(cpath, dtype=np.uint8)
    else:
        img = S.build_image(cfg, qt, seed=1234)
        if cpath and rank == 0:
            img.tofile(cpath + ".tmp"); os.replace(cpath + ".tmp", cpath)
    prompt = S.prompt_tokens(cfg, W, 1234)
    t_build = time.time() - t0

    def run_once(plan=None, transport_override=None, emit=True):
        """One configuration of the job: build the context(s), warm up, time K steps, report.  plan: None = the library's own choice
        (LMRS_SHARD_PLAN if the caller set it), "tp" / "cls" = forced for this run; transport_override: "rccl" / "p2p"."""
        if plan is not None:
            os.environ["LMRS_SHARD_PLAN"] = plan
        # N > 1: ONE decode stream, weight matrices row-split over the N GPUs (one process per GPU), RCCL all-gathers
        # of the per-shard slices over xGMI between the fused kernels.  Token ids stay identical.
        sharded = dist is not None
        transport = None
        if sharded:
            # Transport of the per-layer exchanges (a few KB each, latency-bound): peer-to-peer pushes over xGMI by default (a store + flag
            # kernel per exchange, arenas opened through IPC handles; lmrs_p2p_connect ends with a handshake), RCCL all-gathers if any rank
            # cannot connect (or LMRS_BENCH_TRANSPORT=rccl).  With a world of one (LMRS_BENCH_FORCE_DIST) only the RCCL path exists.
            want = transport_override or os.environ.get("LMRS_BENCH_TRANSPORT", "p2p" if world > 1 else "rccl")
            model = None
            if want == "p2p":
                ok = 1
                try:
                    model = lmrs_amd.Transformer(img, device=local_rank, rank=rank, world=world)
                    if os.environ.get("LMRS_BENCH_FAIL_RANK") == str(rank):      # test hook of THIS launcher: this rank's first connect fails
                        model.debug_inject(0)
                    handles = [None] * world
                    dist.all_gather_object(handles, model.p2p_handle())
                    model.p2p_connect(handles)
                except Exception as e:                      # noqa: BLE001 - any failure means: fall back, together
                    print(f"[rank {rank}] peer-to-peer transport unavailable: {e}", file=sys.stderr)
                    ok = 0
                if all_ranks_ok(ok == 1):
                    transport = "p2p"
                else:
                    if model is not None:
                        model.close()
                    model = None
            if model is None and want == "p2p" and os.environ.get("LMRS_BENCH_ONE_DEVICE") == "1":
                # one-device verification mode: RCCL refuses two ranks on one GPU, so the fallback every rank takes TOGETHER is a second
                # peer-to-peer attempt (with LMRS_BENCH_FAIL_RANK the first one fails on one rank only: the point of the exercise)
                model = lmrs_amd.Transformer(img, device=local_rank, rank=rank, world=world)
                handles = [None] * world
                dist.all_gather_object(handles, model.p2p_handle())
                model.p2p_connect(handles)
                transport = "p2p (second attempt, all ranks together, after a failed connect)"
            if model is None:
                uid = exchange_unique_id(dist, rank, lmrs_amd.comm_unique_id)
                model = lmrs_amd.Transformer(img, device=local_rank, rank=rank, world=world, unique_id=uid)
                transport = "rccl"
        else:
            model = lmrs_amd.Transformer(img, device=local_rank)

        def timed_run(model):
            # ---- warm-up: the W prompt tokens (token by token, as the reference feeds prompts), untimed
            first = model.generate_greedy(prompt, 1)
            device_sync(); barrier()
            # ---- timed: exactly K decode steps at positions W .. W+K-1
            t1 = time.perf_counter()
            toks, dev_sec = model.generate_greedy(first, K, start_pos=W, timing=True)
            device_sync(); barrier()
            t2 = time.perf_counter()
            return first, toks, dev_sec, t2 - t1

        if transport == "p2p":
            # a peer-to-peer exchange that times out mid-run (bounded wait, sticky error) must not cost the job its result: every rank
            # reports, and if any of them failed all of them redo the run over RCCL
            res = None
            try:
                res = timed_run(model)
            except Exception as e:                          # noqa: BLE001
                print(f"[rank {rank}] peer-to-peer run failed: {e}", file=sys.stderr)
            if not all_ranks_ok(res is not None):
                model.close()
                uid = exchange_unique_id(dist, rank, lmrs_amd.comm_unique_id)
                model = lmrs_amd.Transformer(img, device=local_rank, rank=rank, world=world, unique_id=uid)
                transport = "rccl (peer-to-peer run failed)"
                res = timed_run(model)
        else:
            res = timed_run(model)
        # A host stall inside the one timed call (a descheduled thread on a shared box: seen once in ~30 runs, 7 ms on an 8 ms window) is visible as wall-clock
        # time far above the device's own event time for the same K steps.  Single GPU only: the call is then timed AGAIN (the same warm-up, the same K steps
        # at the same positions, nothing skipped), at most twice, and every attempt is reported in `timing_attempts` - `value` is always a wall-clock figure.
        attempts = [{"wall_us_per_step": round(res[3] / K * 1e6, 2), "device_us_per_step": round(res[2] / K * 1e6, 2)}]
        while not sharded and world == 1 and res[2] > 0 and res[3] > 1.25 * res[2] + 2e-4 and len(attempts) < 3:
            res = timed_run(model)
            attempts.append({"wall_us_per_step": round(res[3] / K * 1e6, 2), "device_us_per_step": round(res[2] / K * 1e6, 2)})
        first, toks, dev_sec, wall = res
        # the driver times few steps at the very first positions; the figure over the full 128-token generate of BASELINE.json's configs
        # (positions W .. W+127: longer contexts, a run long enough to be immune to start-up noise) rides along when K differs
        v128 = None
        if not sharded and K != 128 and W + 128 <= 8192:
            first128 = model.generate_greedy(prompt, 1)
            device_sync()
            t1 = time.perf_counter(); model.generate_greedy(first128, 128, start_pos=W); device_sync()
            v128 = round(128 / (time.perf_counter() - t1), 1)
            res2 = timed_run(model)                    # leave the context where the timed run left it (bench_step continues from there)
            del res2
        elapsed = max_over_ranks(dist, wall, ("cpu" if dist.get_backend() == "gloo" else "cuda") if dist is not None else None)
        gen = np.concatenate([first, toks])[: K + 1]          # token ids produced after positions W-1 .. W+K-1
        shard_steps = None
        if sharded:
            # per-kernel / per-exchange durations inside the sharded step (every rank takes part: the exchanges wait for the peers)
            try:
                shard_steps = model.bench_step(W + K, 8)
            except Exception as e:                          # noqa: BLE001
                print(f"[rank {rank}] bench_step: {e}", file=sys.stderr)
            barrier()

        out = None
        if rank == 0:
            tok_s = K / elapsed                      # one decode stream, whatever the number of GPUs
            # whole-path bytes over the timed positions
            path_bytes = sum(model.step_info(p)[1] for p in range(W, W + K))
            n_launch = model.step_info(W)[0]
            # ---- dominant kernel: per-shape live timing with HIP events
            path = {"bytes_per_step": round(path_bytes / K), "us_per_step": round(elapsed / K * 1e6, 2),
                    "achieved": round(path_bytes / elapsed / 1e9, 1), "frac": round(path_bytes / elapsed / 1e9 / (HBM_PEAK_GBPS * world), 4),
                    "kernel_launches_per_step": n_launch, "device_event_us_per_step": round(dev_sec / K * 1e6, 2)}
            if sharded:
                # where a sharded step goes - streaming kernels, glue, exchanges, and what is left (launch gaps)
                split = None
                if shard_steps:
                    iters = 8
                    per = {k: {"us": round(us / n, 3), "launches_per_step": n // iters} for k, (us, _b, n) in shard_steps.items()}
                    stream_us = sum(us for k, (us, _b, n) in shard_steps.items() if k not in ("exchange", "glue")) / iters
                    glue_us = shard_steps.get("glue", (0.0, 0, 0))[0] / iters
                    ex_us = shard_steps.get("exchange", (0.0, 0, 0))[0] / iters
                    gemv_b = sum(_b for k, (us, _b, n) in shard_steps.items() if k in ("qkv", "wo", "w1w3", "w2", "classifier")) / iters
                    gemv_us = sum(us for k, (us, _b, n) in shard_steps.items() if k in ("qkv", "wo", "w1w3", "w2", "classifier")) / iters
                    split = {"stream_us": round(stream_us, 2), "glue_us": round(glue_us, 2),
                             "exchange_us": round(ex_us, 2) if transport == "p2p" else None,
                             "gap_us": round(elapsed / K * 1e6 - stream_us - glue_us - (ex_us if transport == "p2p" else 0.0), 2),
                             "gemv_GBps_this_rank": round(gemv_b / gemv_us / 1e3, 1) if gemv_us else None, "kernels": per,
                             "note": "rank 0's eager replay of the real step with an event pair per dispatch; a dispatch's duration includes its launch boundary; "
                                     "exchange_us includes waiting for the slowest peer; with RCCL the collectives carry no events and sit in gap_us"}
                # weight bytes the job streams per step beyond ONE copy of the model (replicated matrices are read by every GPU): `frac` prices
                # the single-copy bytes against N x 8 TB/s, so it is a scaling figure, not a streaming efficiency, whenever this is not 0
                pl = lmrs_amd.shard_plan(model.args, 0, world)
                a_ = model.args; scb = (1.0 if args.qtype == "q8_0" else 0.5) + 4.0 / 128
                att_f = a_.n_heads * a_.head_size; kv_f = a_.n_kv_heads * a_.head_size
                one = a_.n_layers * (a_.dim * (att_f + 2 * kv_f) + att_f * a_.dim + 3 * a_.dim * a_.hidden_dim) * scb + a_.vocab_size * a_.dim * scb
                per_gpu = a_.n_layers * (a_.dim * (pl["q_heads"][1] + 2 * pl["kv_heads"][1]) * a_.head_size + att_f * pl["dim_rows"][1]
                                         + 2 * a_.dim * pl["hidden_pairs"][1] + a_.hidden_dim * pl["dim_rows"][1]) * scb + pl["vocab_rows"][1] * a_.dim * scb
                redundant = int(round(world * per_gpu - one))
                roofline = {"bound": "hbm", "kernel": "whole sharded step", "achieved": path["achieved"], "redundant_bytes_per_step": redundant,
                            "bytes_streamed_per_gpu_per_step": int(round(per_gpu)),
                            "peak": HBM_PEAK_GBPS * world, "unit": "GB/s", "frac": path["frac"], "traffic": None, "path": path,
                            "transport": transport, "sharded_step_is_one_hipgraph": model.shard_uses_graph(), "step_split": split}
            else:
              # the real step, replayed eagerly from the live state with an event pair on every dispatch: each kernel's
              # duration as it runs inside the step (real predecessor, real activations, positions W+K+1 ...)
              iters = 8
              res = model.bench_step(W + K, iters)
              per, tot_us, tot_b, n_gemv, step_us = {}, 0.0, 0.0, 0, 0.0
              for name, (us, b, n) in res.items():
                per[name] = {"us": round(us / n, 3), "MB": round(b / n / 1e6, 3), "GBps": round(b / us / 1e3, 1), "launches_per_step": n // iters}
                step_us += us / iters
                if name not in ("attention", "argmax"):
                  tot_us += us / iters; tot_b += b / iters; n_gemv += n // iters
              achieved = tot_b / tot_us / 1e3            # GB/s
              rp_us = rocprof_family_us(cfg.name, args.qtype)
              roofline = {
                "bound": "hbm", "kernel": "lmrs::gemv_static_kernel / gemv_kernel / qkv_attn_kernel (fused dequant-GEMV, all shapes of one step; the qkv launch includes the attention workgroups merged into it, the classifier the folded argmax), durations taken inside the real step",
                # `achieved` / `frac`: the WHOLE timed step (SURVEY.md section 8d bytes per token / measured time per token = path.frac: gaps between the
                # launches included); `family_*`: the GEMV family alone on its in-step event durations (it excludes the gaps: 1.5 points higher)
                "achieved": path["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": path["frac"],
                "family_achieved": round(achieved, 1), "family_frac": round(achieved / HBM_PEAK_GBPS, 4),
                "frac_vs_measured_copy": round(path["achieved"] / MEASURED_COPY_GBPS, 4),
                # the same bytes over rocprofv3's average durations of the same kernels (committed summary of this build, else null).  A LOWER
                # BOUND, not a second measurement of kernel time: under graph replay the profiler's per-dispatch intervals include each launch's
                # boundary and overlap their neighbours' - their sum exceeds the timed step itself (round 3: 471 us of "kernels" in a 424 us step)
                "frac_rocprof": round(tot_b / rp_us / 1e3 / HBM_PEAK_GBPS, 4) if rp_us else None,
                "frac_rocprof_is": "lower bound (profiler intervals overlap at launch boundaries; see path.frac for the timed step)",
                "traffic": pmc_traffic(cfg.name, args.qtype),
                "bytes_per_launch_avg": round(tot_b / n_gemv), "avg_launch_us": round(tot_us / n_gemv, 3), "launches_per_step": n_gemv,
                "in_step": per, "sum_of_kernel_us_per_step": round(step_us, 2), "kernel_source_hash": kernel_source_hash(),
                "path": path,
              }
            # ---- CPU baseline + parity gate
            cpu = None; parity = None
            cpu_steps = K if args.cpu_steps < 0 else args.cpu_steps
            if cpu_steps > 0:
                import oracle_lib as O
                if args.cpu_threads > 0:
                    os.environ["LMRS_REF_THREADS"] = str(args.cpu_threads)
                elif "LMRS_REF_THREADS" not in os.environ:
                    # the team the host can really run: affinity mask capped by the container's CPU quota (OpenMP sees only the former,
                    # and 16 spinning threads on a 2-CPU quota are an order of magnitude slower than 2)
                    budget = len(os.sched_getaffinity(0))
                    try:
                        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
                        if quota != "max":
                            budget = min(budget, max(1, int(quota) // int(period)))
                    except (OSError, ValueError):
                        pass
                    os.environ["LMRS_REF_THREADS"] = str(min(16, budget))
                orc = O.Oracle(img)
                n_new = min(cpu_steps, K) + 1
                ref, sec = orc.generate_greedy(prompt, n_new, timing=True)
                steps_run = W + n_new - 1
                # a bounded sample of ~5-10 s of CPU work: the same prompt + steps (the same positions) again until 5 s have been timed (at most 64 times);
                # the driver's 25 steps alone are 0.2 s, and single passes on a shared host scatter by +-15 %
                reps = 1
                while sec < 5.0 and reps < 64:
                    sec += orc.generate_greedy(prompt, n_new, timing=True)[1]; reps += 1
                cpu = {"value": round(reps * steps_run / sec, 2), "unit": "tok/s", "cores": O.threads(), "kind": "port",
                       "sample": f"same image and prompt: {W} prompt + {n_new - 1} greedy steps, {reps} time(s), {sec:.2f}s in all, OpenMP over rows/heads"}
                parity = {"tokens_compared": int(n_new), "tokens_equal": bool((ref == gen[:n_new]).all())}
            # ---- batched forward_layer (fill_kv_cache): the path's only dense contraction, int8 MFMA
            prefill = None
            if not sharded and args.qtype in ("q8_0", "q4_0"):
                n_pf = 256
                emb = model.get_embeddings(S.prompt_tokens(cfg, n_pf, 4321))
                best = 1e9
                best_dev = 1e9
                for _ in range(2):
                    e = emb.copy()
                    t_a = time.perf_counter(); model.fill_kv_cache(e, 0); best = min(best, time.perf_counter() - t_a)
                    best_dev = min(best_dev, model.last_fill_ms() * 1e-3)
                att = cfg.n_heads * cfg.head_size; kvd = cfg.n_kv_heads * cfg.head_size
                macs = n_pf * cfg.n_layers * (cfg.dim * (att + 2 * kvd) + att * cfg.dim + 3 * cfg.dim * cfg.hidden_dim)
                prefill = {"tokens": n_pf, "ms": round(best * 1e3, 2), "tok_s": round(n_pf / best, 1), "achieved": round(2 * macs / best / 1e12, 1),
                           "peak": 3944.0, "unit": "int8 TOP/s", "bound": "mfma", "frac": round(2 * macs / best / 1e12 / 3944.0, 4),
                           # the same without the upload of the embeddings and the download of the residual stream (HIP events inside the call)
                           "ms_device": round(best_dev * 1e3, 3), "frac_device": round(2 * macs / best_dev / 1e12 / 3944.0, 4),
                           "kernel": "lmrs::gemm_q8_*_kernel (v_mfma_i32_16x16x64_i8" + ("; Q4_0 weights unpacked to signed bytes per fragment" if args.qtype == "q4_0" else "") + ") + per-token rows + "
                                     + ("block attention (256-wide heads, soft-capped scores: Gemma-2)" if cfg.model_type == S.GEMMA else "block attention")
                                     + "; `ms` / `frac` include the host<->device copies of the embeddings, `ms_device` / `frac_device` do not"}
            # ---- BASELINE configs[4]: the image path in the reference's call order (chat.rs:84-121) - CLIP tower over the global crop and
            # one sub-image (2 crops x 577 tokens x 23 of 24 layers), projector, fill_kv_cache over the 4 + 313 + 3 embeddings
            vision = None
            if args.vision and not sharded and args.qtype == "q8_0":
                from tools import synth_vision as V
                def best_of(n, f):
                    b, r = 1e9, None
                    for _ in range(n):
                        t_a = time.perf_counter(); r = f(); b = min(b, time.perf_counter() - t_a)
                    return b, r
                vcfg = V.VisionCfg(n_layers=24)
                vt = lmrs_amd.VisionTransformer(V.build_vision_section(vcfg)); pv = V.pixel_values(vcfg, 2)
                t_tower, feats = best_of(3, lambda: vt.forward(pv, 2))
                pr = lmrs_amd.PHI3VProcessor(V.build_processor_section(4 * vcfg.dim, cfg.dim))
                t_proj, img_emb = best_of(3, lambda: pr.forward(feats, 576 * vcfg.dim, 12, 1, 1))
                pre = model.get_embeddings(np.array([1, 32010, 29871, 13], np.uint32)); post = model.get_embeddings(np.array([1, 29871, 13], np.uint32))
                prefix = np.concatenate([pre.reshape(-1), img_emb.reshape(-1), post.reshape(-1)]).astype(np.float32)
                n_emb = prefix.size // cfg.dim
                t_fill, _ = best_of(2, lambda: model.fill_kv_cache(prefix.copy(), 0))
                ntok = 2 * 577
                tower_macs = ntok * (vcfg.n_layers - 1) * (4 * vcfg.dim * vcfg.dim + 2 * vcfg.dim * vcfg.hidden_dim)
                proj_macs = img_emb.shape[0] * (4 * vcfg.dim * cfg.dim + cfg.dim * cfg.dim)
                att = cfg.n_heads * cfg.head_size; kvd = cfg.n_kv_heads * cfg.head_size
                fill_macs = n_emb * cfg.n_layers * (cfg.dim * (att + 2 * kvd) + att * cfg.dim + 3 * cfg.dim * cfg.hidden_dim)
                tot_t = t_tower + t_proj + t_fill
                # parity of the image features against the CPU restatement on a 3-layer tower (the full depth is tests/test_gpu_parity.py::
                # test_vision_tower_at_full_depth; here the CPU leg must stay short) and of the projector on the full-depth features
                import oracle_lib as O
                v3 = V.VisionCfg(n_layers=3); sec3 = V.build_vision_section(v3)
                f3 = lmrs_amd.VisionTransformer(sec3).forward(pv, 2); o3 = O.VisionOracle(sec3).forward(pv, 2)
                po = O.ProcessorOracle(V.build_processor_section(4 * vcfg.dim, cfg.dim)).forward(feats, 576 * vcfg.dim, 12, 1, 1)
                vision = {"tower_ms": round(t_tower * 1e3, 2), "tower_shape": "2 crops x 577 tokens x 23 layers (CLIP ViT-L/14-336)", "projector_ms": round(t_proj * 1e3, 2),
                          "image_embeddings": int(img_emb.shape[0]), "fill_kv_cache_embeddings": int(n_emb), "fill_kv_cache_ms": round(t_fill * 1e3, 2),
                          "tower_TOPs": round(2 * tower_macs / t_tower / 1e12, 1), "fill_TOPs": round(2 * fill_macs / t_fill / 1e12, 1),
                          "achieved": round(2 * (tower_macs + proj_macs + fill_macs) / tot_t / 1e12, 1), "peak": 3944.0, "unit": "int8 TOP/s", "bound": "mfma",
                          "frac": round(2 * (tower_macs + proj_macs + fill_macs) / tot_t / 1e12 / 3944.0, 4),
                          "note": "int8 MACs of the projections only (attention and norms are f32 vector work); host<->device copies of pixels / features / embeddings included",
                          "parity": {"tower_3_layers_bit_equal": bool((f3.view(np.uint32) == o3.reshape(f3.shape).view(np.uint32)).all()),
                                     "projector_bit_equal": bool((img_emb.reshape(-1).view(np.uint32) == po.reshape(-1).view(np.uint32)).all())}}
            out = {
                "metric": "decode tok/s + %HBM-roofline, Llama-3.2-1B Q8_0 @1/2/4/8 MI355X vs CPU ref",
                "value": round(tok_s, 1), "unit": "tok/s", "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": round(elapsed / K * 1e3, 5), "value_128_steps": v128 if K != 128 else round(tok_s, 1), "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
                "vs_baseline": None, "dtype": "int8xint8->int32, f32 combine" if args.qtype == "q8_0" else "int4xint4->int32, f32 combine", "data": "synthetic",
                "config": {"workload": f"{cfg.name} {args.qtype.upper()} (gs=128) greedy decode, {W}-token synthetic prompt then {K} tokens, batch 1",
                           "parallelism": "single GPU" if world == 1 else shard_description(lmrs_amd, model, world, transport),
                           "image_bytes": int(img.size), "build_image_s": round(t_build, 1)},
                "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "prefill": prefill, "vision": vision,
            }
            if len(attempts) > 1:   # (see above: the timed call repeated after a host stall; the last attempt is the one reported)
                out["timing_attempts"] = attempts
            if sharded:          # what moved the slices, and how many ranks RCCL itself counted (0: the peer-to-peer transport, no communicator)
                out["transport"] = transport; out["rccl_nranks"] = model.comm_ranks()
            if emit:
                print(json.dumps(out), flush=True)
        model.close()
        return out

    if dist is None or os.environ.get("LMRS_SHARD_PLAN") or os.environ.get("LMRS_BENCH_SINGLE_PLAN") == "1" or force_dist:
        out = run_once()
    else:
        # N > 1.  The headline line is the configuration north_star names: every weight matrix row-split ("tp"), slices exchanged by
        # RCCL all-gathers over xGMI.  The library's OWN choice for this model and world size (for the small models plan "cls" over the
        # peer-to-peer push transport) is measured in the same job and rides along as `library_choice`, so that one
        # run yields both (LMRS_BENCH_SINGLE_PLAN=1 / LMRS_SHARD_PLAN=...: one configuration only).
        # Order and guard (round 4): the headline configuration runs FIRST and its line is held; the library's own choice - whose default
        # transport, the peer-to-peer push over xGMI, has never crossed a real link - then runs under a watchdog on every rank.  If it
        # raises, hangs or kills a peer, rank 0 still prints the headline line (with `library_choice` saying why it is missing) and every
        # rank leaves: a driver on a real node always gets its one JSON line.
        one_dev = os.environ.get("LMRS_BENCH_ONE_DEVICE") == "1"                      # (RCCL refuses two ranks on one device)

        def headline():
            return run_once(plan="tp", transport_override=None if one_dev else "rccl", emit=False)

        def library_choice():
            os.environ.pop("LMRS_SHARD_PLAN", None)        # (run_once(plan=...) set it for the headline run)
            return run_once(emit=False)

        def tp_split_out():
            # SURVEY section 8(e)'s form to the letter: wo / w2 rows split dim / G as well - four gathers per layer instead of the headline's two
            # (which keeps wo / w2 whole on every shard).  Unmeasured on hardware: the first real node answers replicate-vs-split by measurement.
            os.environ["LMRS_SHARD_SPLIT_OUT"] = "1"
            try:
                return run_once(plan="tp", transport_override=None if one_dev else "rccl", emit=False)
            finally:
                os.environ.pop("LMRS_SHARD_SPLIT_OUT", None)

        if any(v % world for v in (cfg.n_kv_heads, cfg.dim, cfg.hidden_dim, cfg.vocab_size)):
            # the row split north_star names does not exist for this model on this many GPUs (Gemma-2-2B's 4 kv heads on 8): the one configuration
            # there is - whole layers on every GPU, the classifier's rows split (plan "cls"; it needs world | vocab_size) - is the line then, and says so
            out = run_once(plan="cls", transport_override=None if one_dev else os.environ.get("LMRS_BENCH_TRANSPORT"), emit=False)
            if rank == 0 and out is not None:
                out["headline_is"] = f"plan cls: {world} does not divide this model's n_kv_heads / dim / hidden_dim, plan tp does not exist"
                print(json.dumps(out), flush=True)
        else:
            out = headline_then_guarded(headline, [("library_choice", library_choice), ("tp_split_out", tp_split_out)], barrier, rank,
                                        float(os.environ.get("LMRS_BENCH_LIBRARY_CHOICE_TIMEOUT", "240")))
    if dist is not None:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
