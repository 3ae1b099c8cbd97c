import os, sys, time, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def rank_main(rank, world, path, q_in, q_out):
    t0 = time.time()
    def log(*a):
        print(f"[r{rank} +{time.time()-t0:5.1f}s]", *a, flush=True)
    try:
        os.environ["LMRS_P2P_TIMEOUT_MS"] = "1000"; os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np, lmrs_amd
        from tools import synth_lmrs as S
        img = np.fromfile(path, np.uint8)
        log("creating")
        m = lmrs_amd.Transformer(img, device=0, rank=rank, world=world)
        log("created; exporting handle")
        h = m.p2p_handle(); log("handle ok")
        q_out.put((rank, h))
        hs = q_in.get(timeout=60); log("got handles; connecting")
        m.p2p_connect(hs); log("connected; graph =", m.shard_uses_graph())
        t = m.generate_greedy(S.prompt_tokens("mini-llama", 3, 45), 5); log("generated", t)
        q_out.put((rank, "done"))
    except Exception as e:
        import traceback; log("ERROR", traceback.format_exc()); q_out.put((rank, "error"))
if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    from tools import synth_lmrs as S
    img = S.build_image("mini-llama", S.Q8_0, seed=43); path = "/tmp/m.lmrs"; img.tofile(path)
    ctx = mp.get_context("spawn"); q_out = ctx.Queue(); q_in = [ctx.Queue() for _ in range(2)]
    ps = [ctx.Process(target=rank_main, args=(r, 2, path, q_in[r], q_out)) for r in range(2)]
    for p in ps: p.start()
    hs = {}
    for _ in range(2):
        r, h = q_out.get(timeout=50); hs[r] = h
    if all(isinstance(v, bytes) for v in hs.values()):
        for r in range(2): q_in[r].put([hs[0], hs[1]])
        for _ in range(2): print("main:", q_out.get(timeout=50), flush=True)
    for p in ps: p.join(5)
    for p in ps:
        if p.is_alive(): p.kill()
