"""Does hipGraph capture of the per-token launches pay?  Eager launches vs graph replay of the LLM step, the RF sampler and the
semantic-decoder step at a few image counts (full 16B-A3B shapes)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def graphed(fn):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    return gr.replay
for B in [int(v) for v in (sys.argv[1:] or ["1", "4", "16"])]:
    R = 2 * B
    args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=B, cfg_rows=2, weights="bf16")
    cfg, dec, rf, tok = bench.build_models(args, dev, 0)
    x = torch.randn(B, cfg.hidden_size, device=dev, generator=g)
    seq = torch.arange(R, dtype=torch.int32, device=dev)
    slot = torch.full((R,), 168, dtype=torch.int32, device=dev)
    ln = slot + 1
    out = torch.empty(R, cfg.hidden_size, device=dev)
    km = torch.ones(R, dec.t_max, dtype=torch.uint8, device=dev)
    f_llm = lambda: dec.step(x, seq, slot, slot, ln, km, None, out=out, rows=R, x_row_div=2)
    hid = torch.randn(R, cfg.hidden_size, device=dev, generator=g)
    noise = torch.randn(B, 32, device=dev, generator=g)
    lat = torch.empty(B, 32, device=dev)
    f_rf = lambda: rf.sample(hid, noise, n_images=B, out=lat)
    st = tok.new_decode_state(n_seq=B, t_max=256)
    st.length = 100; st.row_slot.fill_(100); st.row_len.fill_(101)
    emb = torch.empty(B, cfg.hidden_size, device=dev); sem = torch.empty(B, tok.feature_dim, device=dev)
    f_sem = lambda: tok.decode_step(lat, st, sem_out=sem, embed_out=emb)
    for name, f in (("LLM step", f_llm), ("RF sample", f_rf), ("semdec step", f_sem)):
        t_e = ev(f)
        try:
            r = graphed(f)
            t_g = ev(r)
        except Exception as ex:
            print(f"images {B}: {name}: capture failed: {ex}", flush=True)
            continue
        print(f"images {B:3d}: {name:12s} eager {t_e:7.3f} ms   graph {t_g:7.3f} ms   ({t_g / t_e:.3f}x)", flush=True)
    del dec, rf, tok, st
    torch.cuda.empty_cache()
