"""Per-component time of one lock-step token on the wide route at full 16B-A3B shapes: LLM step, RF sampler,
semantic-decoder step (HIP events on the current stream)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--images", type=int, default=512); ap.add_argument("--layers", type=int, default=None)
a = ap.parse_args()
args = argparse.Namespace(tiny=False, tokens=256, layers=a.layers, prompt_len=40, images=a.images, cfg_rows=2)
dev = torch.device("cuda", 0)
cfg, dec, rf, tok = bench.build_models(args, dev, 0)
B, R = a.images, 2 * a.images
g = torch.Generator(device=dev).manual_seed(0)
def ev(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
x = torch.randn(B, cfg.hidden_size, device=dev, generator=g)
seq = torch.arange(R, dtype=torch.int32, device=dev)
for T in (48, 168, 290):
    slot = torch.full((R,), T, dtype=torch.int32, device=dev)
    out = torch.empty(R, cfg.hidden_size, device=dev)
    km = torch.ones(R, dec.t_max, dtype=torch.uint8, device=dev)
    t = ev(lambda: dec.step(x, seq, slot, slot, slot + 1, km, None, out=out, rows=R, x_row_div=2))
    print(f"LLM step, {R} rows, cache {T}: {t:.2f} ms", flush=True)
hid = torch.randn(R, cfg.hidden_size, device=dev, generator=g)
noise = torch.randn(B, 32, device=dev, generator=g)
lat = torch.empty(B, 32, device=dev)
print(f"RF sample, {R} rows: {ev(lambda: rf.sample(hid, noise, n_images=B, out=lat)):.2f} ms", flush=True)
st = tok.new_decode_state(n_seq=B, t_max=256)
emb = torch.empty(B, cfg.hidden_size, device=dev); sem = torch.empty(B, tok.feature_dim, device=dev)
def sd():
    st.length = 100; st.row_slot.fill_(100); st.row_len.fill_(101)
    tok.decode_step(lat, st, sem_out=sem, embed_out=emb)
print(f"semdec step, {B} rows, cache 100: {ev(sd):.2f} ms", flush=True)
