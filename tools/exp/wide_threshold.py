"""Where the wide route (gemm256 on hi/lo operands) starts to beat the <= 64-row weight-streaming route, per stage, at full
16B-A3B shapes: ms per call of the LLM step, the RF sampler and the semantic-decoder step for 8..64 rows, both routes."""
import sys, os, argparse, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.devlib  # noqa: F401  (A/B hooks live in libmingnative_dev.so)
import bench
from ming_univision_amd._lib import lib
L = lib()
L.mn_wide_tune.argtypes = [ctypes.c_int] * 3; L.mn_wide_tune.restype = None
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
def ev(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for route, mins in (("stream", 65), ("wide", 2)):
    L.mn_wide_tune(mins, mins, mins)
    for B in (4, 8, 12, 16, 24, 32):
        R = 2 * B
        args = argparse.Namespace(tiny=False, tokens=256, layers=None, prompt_len=40, images=B, cfg_rows=2)
        cfg, dec, rf, tok = bench.build_models(args, dev, 0)
        x = torch.randn(B, cfg.hidden_size, device=dev, generator=g)
        seq = torch.arange(R, dtype=torch.int32, device=dev)
        slot = torch.full((R,), 168, dtype=torch.int32, device=dev)
        out = torch.empty(R, cfg.hidden_size, device=dev)
        km = torch.ones(R, dec.t_max, dtype=torch.uint8, device=dev)
        t_llm = ev(lambda: dec.step(x, seq, slot, slot, slot + 1, km, None, out=out, rows=R, x_row_div=2))
        hid = torch.randn(R, cfg.hidden_size, device=dev, generator=g)
        noise = torch.randn(B, 32, device=dev, generator=g)
        lat = torch.empty(B, 32, device=dev)
        t_rf = ev(lambda: rf.sample(hid, noise, n_images=B, out=lat))
        st = tok.new_decode_state(n_seq=B, t_max=256)
        emb = torch.empty(B, cfg.hidden_size, device=dev); sem = torch.empty(B, tok.feature_dim, device=dev)
        def sd():
            st.length = 100; st.row_slot.fill_(100); st.row_len.fill_(101)
            tok.decode_step(lat, st, sem_out=sem, embed_out=emb)
        t_sem = ev(sd)
        print(f"{route:6s} images {B:3d}: LLM({R} rows) {t_llm:6.2f} ms  RF({R} rows) {t_rf:6.2f} ms  semdec({B} rows) {t_sem:5.2f} ms  "
              f"sum {t_llm + t_rf + t_sem:6.2f} ms -> {B / (t_llm + t_rf + t_sem) * 1e3:6.0f} tok/s", flush=True)
        del dec, rf, tok, st
        torch.cuda.empty_cache()
