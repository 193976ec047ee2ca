"""Shape sweep of the one-row decode step's MoE launches (round 5: moe_gate_up.hip — router + selected experts' gate/up in ONE launch,
every workgroup routing its row itself — and moe_down.hip, in bf16, e4m3, int8 and NF4).  The full-width tests run them at the 16B-A3B
shape only (hidden 2048, 64 + 2 experts, top-6); first contact with another checkpoint must not mis-launch: hidden in {512, 1024, 1536,
2048} (1 .. 4 K chunks per lane; NF4's blocked K map at 1024 / 2048), 8 / 16 / 64 experts, top-k 1 / 2 / 6, 0 .. 2 shared experts, expert
widths that are not multiples of the 6 hidden units a wave owns.  Each case: three 1-row decode steps on a random KV arena against
`oracle/bailing_ref` fed the model's own (de-quantised) weights, 1e-3."""
import pytest
import torch

from ming_univision_amd import configuration as C
from tests.util import llm_sd, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-3
CASES = [  # hidden, experts, top_k, shared, moe_inter, weights
    (512, 8, 1, 0, 64, "bf16"), (512, 16, 2, 1, 200, "bf16"), (1024, 16, 2, 2, 192, "bf16"), (1536, 64, 6, 2, 64, "bf16"),
    (2048, 64, 6, 2, 136, "bf16"), (1024, 16, 2, 1, 192, "fp8"), (2048, 16, 6, 2, 128, "fp8"), (1024, 16, 2, 1, 192, "int8"),
    (1536, 8, 2, 0, 64, "int8"), (1024, 16, 2, 1, 192, "int4"), (2048, 64, 6, 2, 128, "int4"), (512, 8, 2, 1, 128, "int4"),
]


@pytest.mark.parametrize("hidden,experts,top_k,shared,inter,weights", CASES)
def test_one_row_steps_vs_oracle(hidden, experts, top_k, shared, inter, weights):
    from oracle import bailing_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(hidden_size=hidden, num_attention_heads=hidden // 128, num_key_value_heads=max(1, hidden // 512), head_dim=128,
             num_experts=experts, num_experts_per_tok=top_k, num_shared_experts=shared, moe_intermediate_size=inter,
             intermediate_size=inter * 2, vocab_size=512, num_hidden_layers=2, num_image_tokens_for_gen=3, image_start_token=500,
             pad_token_id=0)
    rf_cfg = dict(diffloss_w=64, diffloss_d=2, num_sampling_steps="4", gen_method="flow_matching_swiglu-4")
    sd = llm_sd(d, rf_cfg, 23)
    cfg = C.BailingMoeConfig(**d)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    t_max, n = 8, 4
    dec = BailingMoeDecoder.from_state_dict(cfg, {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}, t_max=t_max, n_seq=1,
                                            weights=weights)
    osd = dict(sd)
    if weights != "bf16":
        for k, v in dec.dequantized_state_dict().items():
            osd[k] = v.float().cpu()
    worst = 0.0
    for seed in range(3):
        g = torch.Generator().manual_seed(7 * hidden + seed)
        x = torch.randn(1, hidden, generator=g) * 0.5
        kv = torch.randn(cfg.num_hidden_layers, 1, 2, cfg.num_key_value_heads, t_max, cfg.head_dim, generator=g) * 0.5
        dec.kv_cache.copy_(kv.cuda())
        slot = torch.full((1,), n, dtype=torch.int32, device="cuda")
        out = dec.step(x.cuda(), torch.zeros(1, dtype=torch.int32).cuda(), slot, slot, slot + 1, distinct_sequences=True)
        kvs = [dict(k=kv[l, :, 0, :, :n].clone(), v=kv[l, :, 1, :, :n].clone()) for l in range(cfg.num_hidden_layers)]
        ref = bailing_ref.model_forward(x[:, None], osd, ocfg, torch.ones(1, n + 1, dtype=torch.long), torch.full((1, 1), n, dtype=torch.long), kvs)[:, 0]
        e = rel_err(out, ref)
        worst = max(worst, e)
        assert torch.isfinite(out).all() and e < TOL, (seed, e)
    print(f"hidden {hidden}, {experts} + {shared} experts, top-{top_k}, width {inter}, {weights}: worst of 3 one-row steps {worst:.2e}")


@pytest.mark.parametrize("hidden,experts,top_k,shared,inter,weights",
                         [(1024, 16, 2, 1, 192, "int8"), (1536, 8, 2, 0, 64, "int8"), (2048, 64, 6, 2, 128, "int4"), (1024, 16, 2, 2, 192, "int4"),
                          (2048, 64, 6, 2, 136, "bf16"), (1024, 16, 2, 1, 192, "fp8")])
def test_two_row_steps_vs_oracle(hidden, experts, top_k, shared, inter, weights):
    """Two rows of distinct sequences (the CFG rows of one image) through the decoder chain: int8 / NF4 take the one-launch router +
    gate/up with the attention projection's slabs summed inside it (and inside the down projection's residual) — two workgroup rounds
    reading the same h, which the launch therefore must not update in place; bf16 / e4m3 the pair launches."""
    from oracle import bailing_ref
    from ming_univision_amd.bailing_moe import BailingMoeDecoder
    d = C.BailingMoeConfig.ming_univision_16b_a3b().to_dict()
    d.pop("model_type", None)
    d.update(hidden_size=hidden, num_attention_heads=hidden // 128, num_key_value_heads=max(1, hidden // 512), head_dim=128,
             num_experts=experts, num_experts_per_tok=top_k, num_shared_experts=shared, moe_intermediate_size=inter,
             intermediate_size=inter * 2, vocab_size=512, num_hidden_layers=2, num_image_tokens_for_gen=3, image_start_token=500,
             pad_token_id=0)
    rf_cfg = dict(diffloss_w=64, diffloss_d=2, num_sampling_steps="4", gen_method="flow_matching_swiglu-4")
    sd = llm_sd(d, rf_cfg, 29)
    cfg = C.BailingMoeConfig(**d)
    ocfg = bailing_ref.LLMConfig(**{k: v for k, v in d.items() if k in bailing_ref.LLMConfig.__dataclass_fields__})
    t_max, n, M = 8, 4, 2
    dec = BailingMoeDecoder.from_state_dict(cfg, {k: v.to("cuda", torch.bfloat16).contiguous() for k, v in sd.items()}, t_max=t_max, n_seq=M,
                                            weights=weights)
    osd = dict(sd)
    if weights != "bf16":
        for k, v in dec.dequantized_state_dict().items():
            osd[k] = v.float().cpu()
    worst = 0.0
    for seed in range(4):
        g = torch.Generator().manual_seed(11 * hidden + seed)
        x = torch.randn(M, hidden, generator=g) * 0.5
        kv = torch.randn(cfg.num_hidden_layers, M, 2, cfg.num_key_value_heads, t_max, cfg.head_dim, generator=g) * 0.5
        dec.kv_cache.copy_(kv.cuda())
        slot = torch.full((M,), n, dtype=torch.int32, device="cuda")
        outs = [dec.step(x.cuda(), torch.arange(M, dtype=torch.int32).cuda(), slot, slot, slot + 1, distinct_sequences=True).clone() for _ in range(2)]
        assert torch.equal(outs[0], outs[1])                     # (a launch that updated h under its own later workgroups would not repeat)
        kvs = [dict(k=kv[l, :, 0, :, :n].clone(), v=kv[l, :, 1, :, :n].clone()) for l in range(cfg.num_hidden_layers)]
        ref = bailing_ref.model_forward(x[:, None], osd, ocfg, torch.ones(M, n + 1, dtype=torch.long), torch.full((M, 1), n, dtype=torch.long), kvs)[:, 0]
        e = rel_err(outs[0], ref)
        worst = max(worst, e)
        assert e < TOL, (seed, e)
    print(f"2 rows, hidden {hidden}, {experts} + {shared} experts, top-{top_k}, width {inter}, {weights}: worst of 4 steps {worst:.2e}")
