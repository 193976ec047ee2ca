"""Per-rank compute time of the TP = 8 decode path at the reference's call shape (one image, 2 CFG rows), measured on ONE GPU:
rank 0's shard of the full 28-layer 16B-A3B stack and of the RF head, every arrival flag pre-raised (the waits fall through), so
the time is what a rank spends in its own kernels per visual token — the xGMI hop of each of the 56 + 192 all-reduces comes on top
(unmeasured here: needs the 8-GPU node).  Prints ms per LLM step / RF sampler / semantic-decoder step and the implied bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ming_univision_amd import configuration as C
from ming_univision_amd.mingtok import MingTok
from ming_univision_amd.rf_head import RectifiedFlowHead
from ming_univision_amd.synth import synth_tensor
from ming_univision_amd.tp import TpCommunicator, TpDecoderShard, TpRfShard

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda")
cfg = C.BailingMoeConfig.ming_univision_16b_a3b()
shard = TpDecoderShard.synthetic(cfg, dev, 0, world, seed=0, t_max=320, n_seq=rows)
rf_cfg = dict(C.DEFAULT_VISHEAD_DIFFLOSS)
rf_sd = {k: synth_tensor(k, s, 0, dev, torch.bfloat16) for k, s in C.llm_param_shapes(cfg, rf_cfg, 32).items()
         if k.startswith("vis_head") or k.startswith("diffloss")}
rf = RectifiedFlowHead(rf_sd, cfg.hidden_size, rf_cfg)
rfs = TpRfShard(rf, 0, world)
comm = TpCommunicator.simulated(world, rows_cap=max(8, rows), width=rf.w)[0]
for t in comm._keep[1]:
    t.fill_(1 << 30)                                   # every arrival flag far ahead of any epoch: waits fall through
x = torch.randn(rows, cfg.hidden_size, device=dev)
seq = torch.arange(rows, dtype=torch.int32, device=dev)
slot = torch.full((rows,), 200, dtype=torch.int32, device=dev)
noise = torch.randn(rows // 2 if rows % 2 == 0 else 1, 32, device=dev)
n_img = noise.shape[0]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_llm = timed(lambda: shard.step_tp(comm, x, seq, slot, slot, slot + 1))
hid = torch.randn(rows, cfg.hidden_size, device=dev)
t_rf = timed(lambda: rfs.sample_tp(comm, hid, noise, n_images=n_img))
tok = MingTok(C.MingTokConfig(), device=dev, seed=0)
st = tok.new_decode_state(n_seq=n_img, t_max=300)
lat = torch.randn(n_img, 32, device=dev)
t_sem = timed(lambda: (tok.decode_step(lat, st), setattr(st, "length", 0), st.row_slot.zero_(), st.row_len.fill_(1)))
tot = t_llm + t_rf + t_sem
print("TP=%d rank-0 compute per visual token at %d rows: LLM step %.3f ms (%d launches-with-all-reduce), RF sampler %.3f ms (%d), "
      "semantic decoder (replicated) %.3f ms: total %.3f ms -> <= %.0f visual tokens/s per image stream before xGMI latency"
      % (world, rows, t_llm, 2 * cfg.num_hidden_layers, t_rf, rf.steps * rf.depth, t_sem, tot, 1e3 / tot * n_img))
print("shard weight bytes: decoder stack %.2f GB, RF head %.2f GB per rank" % (
    shard.weight_bytes() / 1e9, sum(t.numel() * 2 for k in ("w12", "w3") for t in rfs.lists[k]) / 1e9))
