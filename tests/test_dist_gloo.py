"""N > 1 host logic on CPU (gloo, world_size 2): the communicator-id exchange and max-over-ranks timing bench.py uses,
and the row-shard plan + all-gather order of the multi-GPU path (include/lmrs_hip.h: lmrs_shard_plan), checked with the
oracle's matmul_q8 on each rank's rows against the unsharded result (bit-exact: every output row is computed whole on
one rank)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import bench
        import lmrs_amd
        import oracle_lib as O
        from tools import synth_lmrs as S

        # 1. bench.py plumbing
        uid = bench.exchange_unique_id(dist, rank, lambda: bytes(range(128)))
        assert uid == bytes(range(128))
        assert bench.max_over_ranks(dist, 1.0 + rank) == float(world)

        # 2. shard plan + gather order on a real projection shape (wo of mini-llama: 2048 x 2048), row-split plan
        os.environ["LMRS_SHARD_PLAN"] = "tp"
        cfg = S.CONFIGS["mini-llama"]
        a = lmrs_amd.TransformerArgs()
        a.dim, a.hidden_dim, a.n_heads, a.n_kv_heads, a.head_size, a.vocab_size = cfg.dim, cfg.hidden_dim, cfg.n_heads, cfg.n_kv_heads, cfg.head_size, cfg.vocab_size
        plan = lmrs_amd.shard_plan(a, rank, world)
        rng = np.random.default_rng(5)                        # same stream on every rank
        n = cfg.dim; o = cfg.hidden_dim                      # a gate projection: rows split like the (gate, up) pairs
        wq = rng.integers(-127, 128, size=o * n, dtype=np.int8); ws = rng.uniform(1e-4, 3e-3, size=o * n // 128).astype(np.float32)
        xq, xs = O.quantize((rng.standard_normal(n) * 2).astype(np.float32))
        r0, cnt = plan["hidden_pairs"]
        mine = O.matmul_q8(xq, xs, wq[r0 * n:(r0 + cnt) * n], ws[r0 * n // 128:(r0 + cnt) * n // 128], n, cnt)
        parts = [torch.zeros(cnt) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine))
        full = torch.cat(parts).numpy()
        ref = O.matmul_q8(xq, xs, wq, ws, n, o)
        assert (full.view(np.uint32) == ref.view(np.uint32)).all()
        # every row / head / pair / vocab row owned exactly once across ranks
        assert plan["dim_rows"] == (0, cfg.dim)              # wo / w2: replicated, every shard computes the whole residual update
        for key, total in (("q_heads", cfg.n_heads), ("kv_heads", cfg.n_kv_heads), ("hidden_pairs", cfg.hidden_dim), ("vocab_rows", cfg.vocab_size)):
            mine_rng = torch.tensor(plan[key]); allr = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allr, mine_rng)
            cover = sorted((int(t[0]), int(t[1])) for t in allr)
            assert cover[0][0] == 0 and all(cover[i][0] + cover[i][1] == cover[i + 1][0] for i in range(world - 1)) and cover[-1][0] + cover[-1][1] == total
        # the fully row-split form (wo / w2 rows split dim / G as well, four gathers per layer): dim_rows partitioned too, and a
        # wo-shaped projection gathered from the shards' row slices is the unsharded one
        os.environ["LMRS_SHARD_SPLIT_OUT"] = "1"
        ps = lmrs_amd.shard_plan(a, rank, world)
        del os.environ["LMRS_SHARD_SPLIT_OUT"]
        assert ps["dim_rows"] == (rank * (cfg.dim // world), cfg.dim // world) and ps["q_heads"] == plan["q_heads"] and ps["hidden_pairs"] == plan["hidden_pairs"]
        d0, dc = ps["dim_rows"]
        wo_q = rng.integers(-127, 128, size=cfg.dim * n, dtype=np.int8); wo_s = rng.uniform(1e-4, 3e-3, size=cfg.dim * n // 128).astype(np.float32)
        mine_o = O.matmul_q8(xq, xs, wo_q[d0 * n:(d0 + dc) * n], wo_s[d0 * n // 128:(d0 + dc) * n // 128], n, dc)
        parts_o = [torch.zeros(dc) for _ in range(world)]
        dist.all_gather(parts_o, torch.from_numpy(mine_o))
        assert (torch.cat(parts_o).numpy().view(np.uint32) == O.matmul_q8(xq, xs, wo_q, wo_s, n, cfg.dim).view(np.uint32)).all()
        # q heads stay with their kv head (kv_mul q heads per kv head)
        kv_mul = cfg.n_heads // cfg.n_kv_heads
        assert plan["q_heads"][0] == plan["kv_heads"][0] * kv_mul and plan["q_heads"][1] == plan["kv_heads"][1] * kv_mul
        # 3. the other plan ("cls", what the library picks by itself for a model this small): whole layers on every shard, the
        #    classifier's rows split - and the gathered argmax partials give the unsharded answer
        os.environ["LMRS_SHARD_PLAN"] = "cls"
        pc = lmrs_amd.shard_plan(a, rank, world)
        assert pc["q_heads"] == (0, cfg.n_heads) and pc["kv_heads"] == (0, cfg.n_kv_heads) and pc["hidden_pairs"] == (0, cfg.hidden_dim) and pc["dim_rows"] == (0, cfg.dim)
        assert pc["vocab_rows"] == (rank * (cfg.vocab_size // world), cfg.vocab_size // world)
        del os.environ["LMRS_SHARD_PLAN"]
        assert lmrs_amd.shard_plan(a, rank, world) == pc          # the default for this geometry
        V0, VL = pc["vocab_rows"]
        logits = rng.standard_normal(cfg.vocab_size).astype(np.float32)              # same on every rank
        loc = int(np.argmax(logits[V0:V0 + VL]))
        mine_best = torch.tensor([float(logits[V0 + loc]), float(V0 + loc)], dtype=torch.float64)
        allb = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allb, mine_best)
        best = max(allb, key=lambda t_: (float(t_[0]), -float(t_[1])))
        assert int(best[1]) == int(np.argmax(logits))
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_rank_gloo_plumbing_and_shard_plan():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs: p.join(30)
    assert all(r[1] == "ok" for r in res), res


def test_shard_plan_rejects_indivisible_worlds(monkeypatch):
    import lmrs_amd
    monkeypatch.setenv("LMRS_SHARD_PLAN", "tp")
    a = lmrs_amd.TransformerArgs()
    a.dim, a.hidden_dim, a.n_heads, a.n_kv_heads, a.head_size, a.vocab_size = 2048, 8192, 32, 8, 64, 128256
    for w in (1, 2, 4, 8):
        assert lmrs_amd.shard_plan(a, w - 1, w)["vocab_rows"][1] == 128256 // w
    with pytest.raises(lmrs_amd.LmrsError):
        lmrs_amd.shard_plan(a, 0, 3)
    with pytest.raises(lmrs_amd.LmrsError):
        lmrs_amd.shard_plan(a, 0, 16)            # 8 kv heads
    monkeypatch.setenv("LMRS_SHARD_PLAN", "cls")                                  # whole layers, classifier split: only the vocabulary counts
    assert lmrs_amd.shard_plan(a, 2, 3)["vocab_rows"] == (2 * 42752, 42752) and lmrs_amd.shard_plan(a, 0, 16)["q_heads"] == (0, 32)
    with pytest.raises(lmrs_amd.LmrsError):
        lmrs_amd.shard_plan(a, 0, 5)
    monkeypatch.delenv("LMRS_SHARD_PLAN")                                         # the default: by the bytes a shard stops reading
    assert lmrs_amd.shard_plan(a, 0, 8)["q_heads"] == (0, 32)                     # Llama-3.2-1B: the classifier only, at any world size
    a.dim, a.hidden_dim, a.n_heads, a.head_size = 4096, 14336, 32, 128           # Llama-3.1-8B: the layers' matrices are split
    assert lmrs_amd.shard_plan(a, 0, 2)["q_heads"] == (0, 16)
