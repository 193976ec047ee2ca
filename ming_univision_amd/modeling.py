"""MingUniVisionForConditionalGeneration — host-side mirror of mingunivision/modeling_bailingmm.py:85-307.

Owns MingTok (`vision`), the Bailing-MoE decoder (`model`), `linear_proj` and the RF head, wires them the
way the reference does (latent_to_sem_func / linear_proj / sem_to_pix_func, :261-263) and keeps the
multi-round state (KV cache + three attention masks, :124-127, 273-299).  The reference delegates
decoding to HF `GenerationMixin.generate`; only greedy decoding with the `<image>` trigger and EOS is
needed on the hot path (generation_config: do_sample=false), so this class runs its own
greedy loop on top of the C-ABI step functions.
"""
import os

import torch

from . import ops
from ._lib import check, current_stream, lib, ptr
from .bailing_moe import BailingMoeDecoder, build_cfg_rows, generate_image, generate_images
from .configuration import MingUniVisionConfig, linear_proj_param_shapes, llm_param_shapes
from .mingtok import MingTok
from .rf_head import RectifiedFlowHead


def tensor_to_pil(image_tensor):
    """modeling_bailing_moe.py:84-90 — [1,3,H,W] in [-1,1] -> PIL."""
    from PIL import Image
    x = (image_tensor[0].float().cpu() * 0.5 + 0.5).clamp(0, 1)
    arr = (x * 255.0).round().to(torch.uint8).permute(1, 2, 0).numpy()
    return Image.fromarray(arr)


# HF generate kwargs that are no-ops on this path at these values (the reference forwards **generate_kwargs to HF generate,
# modeling_bailingmm.py:249-262): accepted silently when the value is the neutral one, TypeError otherwise.
_NEUTRAL_GENERATE_KWARGS = {
    "num_beams": 1, "num_beam_groups": 1, "num_return_sequences": 1, "repetition_penalty": 1.0, "length_penalty": 1.0,
    "no_repeat_ngram_size": 0, "penalty_alpha": None, "typical_p": 1.0, "min_p": None, "min_new_tokens": 0, "min_length": 0,
    "early_stopping": False, "output_scores": False, "output_attentions": False, "output_hidden_states": False,
    "return_dict_in_generate": False, "synced_gpus": False, "streamer": None, "bad_words_ids": None, "logits_processor": None,
    "stopping_criteria": None,
}
_IGNORED_GENERATE_KWARGS = ("pad_token_id", "bos_token_id", "use_cache", "return_tensors")   # change nothing for one unpadded sequence


def filter_generate_kwargs(kw, default_eos, who):
    """-> the set of end-of-sequence ids the decode loop stops at.  Raises TypeError for every kwarg that would change the result."""
    kw = dict(kw)
    eos = kw.pop("eos_token_id", default_eos)
    eos_ids = set(eos) if isinstance(eos, (list, tuple, set)) else {eos}
    for k in _IGNORED_GENERATE_KWARGS:
        kw.pop(k, None)
    bad = {}
    for k, v in kw.items():
        if k in _NEUTRAL_GENERATE_KWARGS and (v == _NEUTRAL_GENERATE_KWARGS[k] or v is None
                                              or (k in ("logits_processor", "stopping_criteria", "bad_words_ids") and not v)):
            continue
        bad[k] = v
    if bad:
        raise TypeError(f"{who}: unsupported arguments {sorted(bad)} (supported beyond the reference's own: do_sample, temperature, "
                        "top_k, top_p, generator, eos_token_id; HF kwargs are accepted only at values that change nothing)")
    return eos_ids


def check_sampling_args(do_sample, temperature, top_k, top_p, who):
    if do_sample and not (temperature > 0 and top_k >= 0 and 0 < top_p <= 1):
        raise ValueError(f"{who}: do_sample needs temperature > 0, top_k >= 0 and 0 < top_p <= 1")
    if do_sample and top_k > ops.SAMPLE_CANDIDATES:
        raise ValueError(f"{who}: top_k = {top_k} exceeds the {ops.SAMPLE_CANDIDATES} candidates the sampler ranks")


class MingUniVisionForConditionalGeneration:
    config_class = MingUniVisionConfig
    BATCH_SEQ0 = 3      # cache sequences 0..2 hold the (up to 3) CFG rows of the multi-round conversation of generate()

    def __init__(self, config: MingUniVisionConfig, state_dict=None, device="cuda", seed=0, t_max=4096, weights="bf16"):
        """state_dict: reference-named tensors (`vision.*`, `model.model.*`, `model.lm_head.*`, `model.vis_head.*`,
        `model.diffloss.*`, `linear_proj.*`); None -> deterministic synthetic weights (synth.py).
        weights="fp8": the experts of the decoder stack and the ResBlock matrices of the RF head are quantised at load to OCP e4m3
        with one scale per output row (mingnative.h section 7) — the counterpart of the reference's weight-only `dtype` modes
        (mingunivisioninfer.py:46-70); everything else, and all arithmetic, stays as in "bf16"."""
        self.weights = weights
        assert config.llm_config is not None
        assert config.vishead_diffloss_config is not None          # modeling_bailingmm.py:118
        self.config = config
        self.device = torch.device(device)
        cfg = config.llm_config
        from .synth import synth_tensor
        from .configuration import MingTokConfig
        tcfg = config.mingtok_config or MingTokConfig()

        def bf(t):
            return t.to(self.device, torch.bfloat16).contiguous()
        D = tcfg.semantic_decoder.get("embed_dim", 1024)
        lp_shapes = linear_proj_param_shapes(D, cfg.hidden_size, config.mlp_depth)
        if state_dict is None:
            lp = {k: synth_tensor(k, s, seed, self.device, torch.bfloat16) for k, s in lp_shapes.items()}
            tok_sd = None
        else:
            lp = {k: bf(state_dict[k]) for k in lp_shapes}
            tok_sd = {k[len("vision."):]: v for k, v in state_dict.items() if k.startswith("vision.")}
        lp = ops.convert_linears(lp, weights)               # int4 / int8: every nn.Linear of the model is converted (_lib.FULL_MODEL)
        self.linear_proj = [(lp[f"linear_proj.{2 * i}.weight"], lp[f"linear_proj.{2 * i}.bias"])
                            for i in range(config.mlp_depth)]
        self.vision = MingTok(tcfg, state_dict=tok_sd, device=self.device, seed=seed, linear_proj=self.linear_proj, weights=weights)
        if state_dict is None:
            self.model = BailingMoeDecoder.synthetic(cfg, self.device, seed=seed, t_max=t_max, n_seq=3, weights=weights)
            shapes = llm_param_shapes(cfg, config.vishead_diffloss_config, self.vision.latent_dim)
            rf_sd = {k: synth_tensor(k, s, seed, self.device, torch.bfloat16) for k, s in shapes.items()
                     if k.startswith("vis_head") or k.startswith("diffloss")}
        else:
            llm_sd = {k[len("model."):]: bf(v) for k, v in state_dict.items() if k.startswith("model.")}
            self.model = BailingMoeDecoder.from_state_dict(cfg, llm_sd, t_max=t_max, n_seq=3, weights=weights)
            rf_sd = {k: v for k, v in llm_sd.items() if k.startswith("vis_head") or k.startswith("diffloss")}
        self.rf = RectifiedFlowHead(rf_sd, cfg.hidden_size, config.vishead_diffloss_config, self.vision.latent_dim, weights=weights)
        self._init_state(seed)

    @classmethod
    def from_parts(cls, config, vision, model, rf, linear_proj, seed=0):
        """Assemble the wrapper around components that already live in HBM (the way the reference's __init__ wires them,
        modeling_bailingmm.py:93-129) — e.g. a second view of the same weights with another KV arena.  `model` is a
        BailingMoeDecoder, `vision` a MingTok built with the same linear_proj list, `rf` a RectifiedFlowHead."""
        self = cls.__new__(cls)
        self.config, self.device = config, model.device
        self.weights = getattr(model, "weights", "bf16")
        self.linear_proj, self.vision, self.model, self.rf = linear_proj, vision, model, rf
        self._init_state(seed)
        return self

    def _init_state(self, seed):
        self.tokenizer = None
        self.mfma_prefill_threshold = 64   # prompts longer than this prefill as GEMMs (prefill_wide / prefill_mfma)
        # Numerics of what feeds the LLM in understanding / editing (MingTok encode + linear_proj, long-prompt prefill):
        #   "fp32" (default): fp32-class — hi/lo GEMMs, fp32 attention; within 1e-3 of the fp32 reference path (north_star);
        #   "bf16": bf16 activations like the reference's own torch.autocast path (modeling_bailingmm.py:132, 243): ~2x faster
        #           prefill, ~1e-2 of the fp32 result (DESIGN.md §2 has both measured).
        self.understanding_precision = "fp32"
        self.decode_chunk = 8              # greedy text tokens per host round trip
        self.noise_generator = torch.Generator(device=self.device)
        self.noise_generator.manual_seed(seed)
        self.reset_inner_state()

    def _warn_if_sampling_truncated(self):
        """After a host sync of a sampled decode: say so (once per call site) when a draw's kept set did not fit the sampler's
        candidates — the only case in which the draw is not HF's warped distribution."""
        probe = getattr(self.model, "sampling_truncated", None) or getattr(getattr(self.model, "full", None), "sampling_truncated", None)
        bits = probe() if probe else 0
        if bits:
            import warnings
            what = []
            if bits & ops.SAMPLE_NUCLEUS_TRUNCATED:
                what.append(f"a top-p nucleus of more than {ops.SAMPLE_CANDIDATES} tokens was cut to the {ops.SAMPLE_CANDIDATES} most likely")
            if bits & ops.SAMPLE_TIES_TRUNCATED:
                what.append("ties at the top-k threshold exceeded the candidate capacity (lowest ids kept)")
            warnings.warn("sampled decode: " + "; ".join(what))

    # ---- state -----------------------------------------------------------------------------------
    def reset_inner_state(self):
        """modeling_bailingmm.py:303-307"""
        self.past_len = 0
        self.past_attention_mask = None
        self.past_text_uncond_attention_mask = None
        self.past_uncond_attention_mask = None

    # ---- vision ----------------------------------------------------------------------------------
    def extract_image_feature(self, pixel_values, grid_thw=None, precision=None):
        """MingTok.forward -> x_norm_patchtokens -> linear_proj (modeling_bailingmm.py:131-138) -> [B*N, H] fp32."""
        precision = precision or self.understanding_precision
        feat = self.vision.forward(pixel_values, precision=precision)["x_norm_patchtokens"]
        x32 = feat.reshape(-1, feat.shape[-1]).contiguous()
        n = len(self.linear_proj)
        if precision == "fp32":
            for i, (w, b) in enumerate(self.linear_proj):               # Linear [GELU Linear]* on hi/lo operands
                a2, _ = ops.norm_act_split(x32, "none", gelu=i > 0)
                x32 = ops.linear_hilo(a2, w, b)
            return x32
        x = ops.f32_to_bf16(x32)
        for i, (w, b) in enumerate(self.linear_proj):
            last = i == n - 1
            x = ops.gemm_bf16(x, w, b, "f32" if last else "bf16_gelu")
        return x

    def prompt_wrap_vision(self, input_ids, inputs_embeds, vision_embeds, image_token_id=None):
        """masked_scatter of image features at `<imagePatch>` positions (modeling_bailingmm.py:152-177).
        input_ids [1, T] (or [T]); inputs_embeds fp32 [T, H] (or [1, T, H]); vision_embeds [N, H] or [B, N, H].
        Returns (inputs_embeds, image_router_mask): the mask is bool, shaped like input_ids, True on image rows; it is None
        when there is nothing to scatter (the reference returns the bare embeddings in that case, :153-154 — its only
        caller, prompt_wrap_navit, never takes that branch; here the result is always a pair).  Raises ValueError when the
        number of `<imagePatch>` tokens and of feature rows differ (:163-166)."""
        if vision_embeds is None or input_ids is None:
            return inputs_embeds, None
        if image_token_id is not None:                                   # the reference stores the override in the config (:159)
            self.config.llm_config.image_patch_token = image_token_id
        patch = self.config.llm_config.image_patch_token
        vision_embeds = vision_embeds.reshape(-1, vision_embeds.shape[-1])
        router_mask = input_ids == patch
        n_tok, n_feat = int(router_mask.sum()), vision_embeds.shape[0]
        if n_tok != n_feat:
            raise ValueError(f"Image features and image tokens do not match: tokens: {n_tok}, features {n_feat}")
        out = inputs_embeds.clone()
        flat = out.reshape(-1, out.shape[-1])
        flat[router_mask.reshape(-1).to(out.device)] = vision_embeds.to(out.device, out.dtype)
        return out, router_mask

    def prompt_wrap_navit(self, input_ids, query_embeds_image=None, query_embeds_video=None, query_embeds_audio=None,
                          query_embeds_audio_lengths=None, placeholder_audio_loc_lens=None, target_embeds=None):
        """modeling_bailingmm.py:179-204: embedding lookup, then the image (or video) features scattered over the
        `<imagePatch>` rows.  Returns the bare embeddings [T, H] fp32 when no modality is given (as the reference does),
        else (inputs_embeds, image_mask, audio_mask).  Audio is outside the hot path."""
        if query_embeds_audio is not None:
            raise NotImplementedError("audio inputs are outside the hot path (SURVEY.md a27)")
        ids = input_ids.to(self.device)
        inputs_embeds = self.model.embed(ids.reshape(-1))
        if query_embeds_image is None and query_embeds_video is None and target_embeds is None:
            return inputs_embeds
        image_mask = None
        if query_embeds_image is not None:
            inputs_embeds, image_mask = self.prompt_wrap_vision(ids, inputs_embeds, query_embeds_image)
        if query_embeds_video is not None:
            inputs_embeds, image_mask = self.prompt_wrap_vision(ids, inputs_embeds, query_embeds_video)
        return inputs_embeds, image_mask, None

    # ---- batched text -> image (extension: the reference generates one image per call, modeling_bailing_moe.py:1865) ------
    @torch.no_grad()
    def generate_image_batch(self, requests, output_image_prefixes=None, forced_first_token=None, n_groups=1, noises=None,
                             image_gen_temperature=1.0, save=True):
        """B independent single-round text->image requests advanced in lock-step (bailing_moe.generate_images): each image is
        what `generate` would produce for its request alone with the same noise.  requests: dicts with `input_ids` [1, T_i],
        `attention_mask`, `uncond_attention_mask`, `text_uncond_attention_mask` (as BailingMMProcessor returns them; prompt
        lengths may differ, the CFG row count must not).  The first generated token of every request must be `<image>` (or be
        forced).  Does not touch the multi-round state.  Returns dict(images [B,3,R,R], files, latents, sem)."""
        cfg, dev = self.config.llm_config, self.device
        B = len(requests)
        ids, ams, uncs, tuncs = [], [], [], []
        one = torch.ones(1, 1, dtype=torch.long)
        for r in requests:
            i = r["input_ids"].reshape(1, -1)
            am = torch.ones_like(i) if r.get("attention_mask") is None else r["attention_mask"].cpu().long()
            unc = am.clone() if r.get("uncond_attention_mask") is None else r["uncond_attention_mask"].cpu().long()
            tunc = am.clone() if r.get("text_uncond_attention_mask") is None else r["text_uncond_attention_mask"].cpu().long()
            ids.append(i.clip(0, cfg.vocab_size - 1)); ams.append(torch.cat((am, one), 1)); uncs.append(unc); tuncs.append(tunc)
        rpi = {build_cfg_rows(a, u, t).shape[0] for a, u, t in zip(ams, uncs, tuncs)}
        if len(rpi) != 1:
            raise ValueError(f"all requests of a batch must have the same number of CFG rows, got {sorted(rpi)}")
        rpi = rpi.pop()
        lens = [int(i.shape[1]) for i in ids]
        n_tok = cfg.num_image_tokens_for_gen
        if max(lens) + n_tok + 2 > self.model.t_max:
            raise ValueError(f"prompt of {max(lens)} tokens + {n_tok} image tokens exceed the KV arena (t_max = {self.model.t_max})")
        s0 = self.BATCH_SEQ0            # the multi-round conversation keeps its cache sequences
        self.model.ensure_sequences(s0 + rpi * B)
        hidden = self.model.prefill_ragged([self.model.embed(i[0].to(dev)) for i in ids], [s0 + rpi * b for b in range(B)])
        first = self.model.greedy(hidden).tolist()
        if forced_first_token is not None:
            first = [int(forced_first_token)] * B
        bad = [b for b, t in enumerate(first) if t != cfg.image_start_token]
        if bad:
            raise ValueError(f"requests {bad} do not start an image (first tokens {[first[b] for b in bad]}); use generate() for them")
        if noises is None:   # the same draws, in the same order, as B successive generate() calls
            noises = torch.stack([torch.randn(n_tok + 1, self.vision.latent_dim, generator=self.noise_generator, device=dev)
                                  for _ in range(B)])
        start = self.model.embed(torch.tensor([cfg.image_start_token], device=dev))
        try:
            out = generate_images(self.model, self.rf, self.vision, start, lens, ams, uncs, tuncs, noises.to(dev),
                                  temperature=image_gen_temperature, text_cfg=3.0, image_cfg=1.1, n_groups=n_groups, seq0=s0)
        finally:
            self.model.release_sequences(s0)         # the batch's cache sequences are dead: give the arena back
        files = []
        if save:
            prefixes = output_image_prefixes or [f"output_{b}" for b in range(B)]
            for b in range(B):
                name = f"{prefixes[b]}.png"
                tensor_to_pil(out["image"][b:b + 1]).save(name)
                files.append(name)
        return dict(images=out["image"], files=files, latents=out["latents"], sem=out["sem"])

    # ---- batched greedy text decoding (extension: image -> text understanding for B conversations in lock-step) -----------
    @torch.no_grad()
    def generate_text_batch(self, requests, max_new_tokens=64, sync_every=8, timings=None, do_sample=False, temperature=1.0, top_k=50,
                            top_p=1.0, generator=None):
        """B independent single-round conversations decoded greedily in lock-step: one pass through the decoder stack per new
        token serves all B sequences (rows = B: the weight-streaming kernels up to 64 sequences, the wide MFMA route with
        grouped-GEMM experts above), lm_head as one GEMM, argmax / embedding lookup / row bookkeeping on the device; the host
        looks at the finished flags every `sync_every` tokens.  requests: dicts with `input_ids` [1, T_i] and optionally
        `pixel_values` / `image_grid_thw` (as BailingMMProcessor returns them; attention masks must be all ones).  Every
        sequence gets the tokens `generate` would give it alone (greedy, up to its first EOS).  Image generation is not
        triggered here (`<image>` is returned as a token).  Does not touch the multi-round state (its cache sequences included).
        Returns a list of B token-id lists (EOS included when reached).  `timings` (a dict, measurement only): synchronises
        after the prefills and at the end and stores `prefill_s` / `decode_s`.  `do_sample` / `temperature` / `top_k` / `top_p` /
        `generator` as in `generate`: new token i of sequence b uses uniform [i, b] of one torch.rand(max_new_tokens, B) draw."""
        check_sampling_args(do_sample, temperature, top_k, top_p, "generate_text_batch")
        sampling = (float(temperature), int(top_k), float(top_p), generator) if do_sample else None
        try:
            return self._generate_text_batch(requests, max_new_tokens, sync_every, timings, sampling)
        finally:
            self.model.release_sequences(self.BATCH_SEQ0)     # the batch's cache sequences are dead: give the arena back

    def _generate_text_batch(self, requests, max_new_tokens, sync_every, timings, sampling=None):
        import time
        cfg, dev = self.config.llm_config, self.device
        B = len(requests)
        t_start = time.perf_counter()
        s0 = self.BATCH_SEQ0                # the multi-round conversation keeps its cache sequences
        self.model.ensure_sequences(s0 + B)
        lens, ids_l, embeds, masks = [], [], [], [None] * B
        for b, r in enumerate(requests):
            ids = r["input_ids"].reshape(1, -1).to(dev).clip(0, cfg.vocab_size - 1)
            am = r.get("attention_mask")
            if am is not None and int(am.sum()) != am.numel():
                raise ValueError("generate_text_batch: attention masks with holes are not supported (use generate)")
            T = ids.shape[1]
            if T + max_new_tokens > self.model.t_max:
                raise ValueError(f"request {b}: {T} prompt tokens + {max_new_tokens} new tokens exceed the KV arena (t_max = {self.model.t_max})")
            ids_l.append(ids)
            lens.append(T)
            embeds.append(self.model.embed(ids[0]))
        # vision tower: requests whose images have the same shape go through MingTok as batches (<= ~8M pixels per pass)
        by_shape = {}
        for b, r in enumerate(requests):
            if r.get("pixel_values") is not None and lens[b] > 1:
                by_shape.setdefault(tuple(r["pixel_values"].shape), []).append(b)
        for shape, members in by_shape.items():
            per = max(1, (8 << 20) // max(1, shape[0] * shape[-1] * shape[-2]))
            for c0 in range(0, len(members), per):
                chunk = members[c0:c0 + per]
                px = torch.cat([requests[b]["pixel_values"].to(dev) for b in chunk], 0)
                feats = self.extract_image_feature(px)
                feats = feats.reshape(len(chunk), -1, feats.shape[-1])
                for j, b in enumerate(chunk):
                    embeds[b], m = self.prompt_wrap_vision(ids_l[b], embeds[b], feats[j])
                    masks[b] = m.reshape(-1)
        # prompts, fp32-class (default): all of them in shared passes of <= max_rows() rows through the stack (decode-path numerics)
        # bf16 regime: long ones on the bf16 MFMA path, stacked up to 8192 tokens per pass; short ones through the decode kernels
        last = [None] * B
        long_ = [b for b in range(B) if lens[b] > self.mfma_prefill_threshold and cfg.head_dim == 128]
        if self.understanding_precision == "fp32" and self.model.max_rows() > 64:
            hs = self.model.prefill_ragged(embeds, [s0 + b for b in range(B)], past=0, image_masks=masks)
            last = [hs[b:b + 1] for b in range(B)]
            long_ = []
        c0 = 0
        while c0 < len(long_):
            c1, tot = c0, 0
            while c1 < len(long_) and (c1 == c0 or tot + lens[long_[c1]] <= 8192):
                tot += lens[long_[c1]]
                c1 += 1
            chunk = long_[c0:c1]
            hs = self.model.prefill_mfma_many([embeds[b] for b in chunk], [s0 + b for b in chunk], past=0,
                                              image_masks=[masks[b] for b in chunk])
            for j, b in enumerate(chunk):
                last[b] = hs[j:j + 1]
            c0 = c1
        for b in range(B):
            if last[b] is None:
                last[b] = self.model.prefill(embeds[b], seq=s0 + b, past=0, image_mask=masks[b])[-1:]
        hidden = torch.cat(last, 0).contiguous()
        if timings is not None:
            torch.cuda.synchronize(dev)
            t_prefill = time.perf_counter()
        seq = torch.arange(s0, s0 + B, dtype=torch.int32, device=dev)
        slot = torch.tensor(lens, dtype=torch.int32, device=dev)
        ln = slot + 1
        finished = torch.zeros(B, dtype=torch.bool, device=dev)
        toks = []
        uniforms = torch.rand(max_new_tokens, B, device=dev, generator=sampling[3]) if sampling else None
        for step in range(max_new_tokens):
            tok = self.model.sample(hidden, uniforms[step], *sampling[:3]) if sampling else self.model.greedy(hidden)
            toks.append(tok)
            finished |= tok == cfg.eos_token_id
            if step + 1 == max_new_tokens:
                break
            if (step + 1) % sync_every == 0 and bool(finished.all()):      # the only host syncs of the loop
                break
            hidden = self.model.step(self.model.embed(tok), seq, slot, slot, ln, distinct_sequences=True)
            check(lib().mn_rows_advance(ptr(slot), ptr(ln), None, B, 1, current_stream()), "mn_rows_advance")
        out = torch.stack(toks, 1).tolist()
        if sampling:
            self._warn_if_sampling_truncated()
        if timings is not None:
            timings.update(prefill_s=t_prefill - t_start, decode_s=time.perf_counter() - t_prefill, steps=len(toks) - 1)
        res = []
        for row in out:
            if cfg.eos_token_id in row:
                row = row[:row.index(cfg.eos_token_id) + 1]
            res.append(row)
        return res

    # ---- generation --------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, uncond_attention_mask=None, text_uncond_attention_mask=None,
                 pixel_values=None, image_grid_thw=None, past_key_values=None, output_image_prefix="output",
                 image_gen_temperature=1.0, image_gen_text_cfg=3.0, image_gen_image_cfg=1.1, max_new_tokens=512,
                 use_cache=True, forced_first_token=None, do_sample=False, temperature=1.0, top_k=50, top_p=1.0,
                 generator=None, image_noises=None, **generate_kwargs):
        """Decode with the `<image>` trigger (modeling_bailingmm.py:206-301; modeling_bailing_moe.py:1769-1796).
        Returns the LongTensor `sequences` [1, T_in + n_new] like HF generate.  `forced_first_token` (extension
        used by benchmarks/tests with random weights) overrides the first generated id; `image_noises` (extension: fp32
        [num_image_tokens_for_gen + 1, latent_dim]) replaces the noise RectifiedFlowLoss.sample would draw for the call's image
        (diff_loss_rf_swiglu.py:117-122: torch.randn per visual token) — how a recorded run of the reference is replayed.

        Text tokens are picked greedily (the checkpoint's generation config: do_sample = false, mingunivision/config.json:30) or,
        with `do_sample=True`, drawn the way HF generate draws them — the kwargs the reference forwards to it (:249-262):
        `temperature`, `top_k` (0 = off; HF's default 50), `top_p` — on the device (mn_sample_logits), from the uniforms of
        `generator` (a torch.Generator on the model's device; None = torch's default CUDA stream).  New token i of a call uses
        the i-th uniform, so a run is reproducible from the generator's seed.  Other HF generate kwargs: the ones that change
        nothing on this path at the value given (`pad_token_id`, `use_cache`, `num_beams=1`, `repetition_penalty=1.0`, …) are
        accepted, `eos_token_id` is honoured, anything that WOULD change the result raises TypeError (`filter_generate_kwargs`):
        there is no silent subset of HF generate here."""
        eos_ids = filter_generate_kwargs(generate_kwargs, self.config.llm_config.eos_token_id, "generate")
        check_sampling_args(do_sample, temperature, top_k, top_p, "generate")
        cfg = self.config.llm_config
        dev = self.device
        assert input_ids.shape[0] == 1, "the reference path is batch-size 1 (modeling_bailing_moe.py:1865)"
        input_ids = input_ids.to(dev)
        T = input_ids.shape[1]
        attention_mask = torch.ones(1, T, dtype=torch.long) if attention_mask is None else attention_mask.cpu()
        unc = attention_mask.clone() if uncond_attention_mask is None else uncond_attention_mask.cpu()
        tunc = attention_mask.clone() if text_uncond_attention_mask is None else text_uncond_attention_mask.cpu()
        if self.past_attention_mask is not None:                         # :231-234
            attention_mask = torch.cat((self.past_attention_mask, attention_mask), dim=1)
            unc = torch.cat((self.past_uncond_attention_mask, unc), dim=1)
            tunc = torch.cat((self.past_text_uncond_attention_mask, tunc), dim=1)
        prompt_mask_len = attention_mask.shape[1]
        ids = input_ids.clip(0, cfg.vocab_size - 1)
        image_mask = None
        if pixel_values is not None and T > 1:                           # :237-248
            feats = self.extract_image_feature(pixel_values.to(dev), image_grid_thw)
            embeds, image_mask, _ = self.prompt_wrap_navit(ids, feats)
            image_mask = image_mask.reshape(-1)
        else:
            embeds = self.model.embed(ids[0])
        past = self.past_len
        n_img_tok = cfg.num_image_tokens_for_gen
        if past + T > self.model.t_max:
            raise ValueError(f"{past} cached + {T} prompt tokens exceed the KV arena (t_max = {self.model.t_max}); "
                             "call reset_inner_state() or build the model with a larger t_max")
        if T > self.mfma_prefill_threshold and self.understanding_precision == "fp32" and self.model.max_rows() > 64:
            # long prompts (image understanding: 256-1024 image tokens), fp32-class: the decode path's numerics at MFMA rate
            hidden = self.model.prefill_wide(embeds, seq=0, past=past, image_mask=image_mask)[-1:]
        elif T > self.mfma_prefill_threshold and self.config.llm_config.head_dim == 128:
            # bf16 activations (the reference's autocast precision): bf16 MFMA prefill with grouped-GEMM MoE, flash attention
            hidden = self.model.prefill_mfma(embeds, seq=0, past=past, image_mask=image_mask)
        else:
            hidden = self.model.prefill(embeds, seq=0, past=past, image_mask=image_mask)[-1:]
        cache_len = past + T
        am = attention_mask
        new_ids = []
        n_img = 0
        one = torch.ones(1, 1, dtype=am.dtype)
        seq0 = torch.zeros(1, dtype=torch.int32, device=dev)
        done = False
        uniforms = torch.rand(max_new_tokens, device=dev, generator=generator) if do_sample else None
        # Greedy decode in chunks of `decode_chunk` tokens: argmax, embedding lookup and the row bookkeeping stay on the device, the
        # host reads the chunk's token ids once (one sync per chunk instead of one per token) and rolls back to the first EOS /
        # `<image>` it finds — greedy decoding is deterministic, the speculative steps behind such a token only wrote cache slots
        # that the next real step overwrites.  Sampling keeps the scheme: new token i always uses uniforms[i], whatever the chunking.
        while not done and len(new_ids) < max_new_tokens:
            # every token of a chunk is fed (cache slot + rotary position cache_len + j) before the host looks at it: the
            # chunk must end inside the arena, also when the conversation stops a few slots short of t_max
            room = self.model.t_max - cache_len
            remaining = max_new_tokens - len(new_ids)
            n = min(self.decode_chunk, remaining)
            if (n if n < remaining else n - 1) > room:                   # tokens this chunk feeds (the call's last token is not fed)
                if room <= 0:
                    raise ValueError(f"the conversation filled the KV arena (t_max = {self.model.t_max}) after {len(new_ids)} new tokens")
                n = room
            slot = torch.tensor([cache_len], dtype=torch.int32, device=dev)
            ln = slot + 1
            toks_dev = []
            for j in range(n):
                if do_sample:
                    i_new = len(new_ids) + j
                    tok_dev = self.model.sample(hidden[0:1], uniforms[i_new:i_new + 1], temperature, top_k, top_p)
                else:
                    tok_dev = self.model.greedy(hidden[0:1])
                if not new_ids and j == 0 and forced_first_token is not None:
                    tok_dev = torch.tensor([int(forced_first_token)], device=dev)
                toks_dev.append(tok_dev)
                if j + 1 < n or len(new_ids) + n < max_new_tokens:       # the last token of the call is never fed
                    hidden = self.model.step(self.model.embed(tok_dev), seq0, slot, slot, ln, distinct_sequences=True)
                    check(lib().mn_rows_advance(ptr(slot), ptr(ln), None, 1, 1, current_stream()), "mn_rows_advance")
            toks = torch.cat(toks_dev).tolist()                          # the chunk's only host sync
            if do_sample:
                self._warn_if_sampling_truncated()
            for j, tok in enumerate(toks):
                new_ids.append(tok)
                if tok in eos_ids or len(new_ids) == max_new_tokens:
                    cache_len += j                                       # tokens 0..j-1 of the chunk were fed
                    done = True
                    break
                if tok == cfg.image_start_token:
                    cache_len += j
                    if am.shape[1] < cache_len:                          # pad the mask over generated tokens
                        am = torch.cat((am, torch.ones(1, cache_len - am.shape[1], dtype=am.dtype)), dim=1)
                    x = self.model.embed(torch.tensor([tok], device=dev))
                    n_tok = n_img_tok
                    if cache_len + n_tok + 1 > self.model.t_max:
                        raise ValueError(f"{cache_len} cached tokens + {n_tok + 1} image slots exceed the KV arena (t_max = {self.model.t_max})")
                    if image_noises is not None:
                        noises = image_noises.to(dev, torch.float32).contiguous()
                        assert noises.shape == (n_tok + 1, self.vision.latent_dim), tuple(noises.shape)
                    else:
                        noises = torch.randn(n_tok + 1, self.vision.latent_dim, generator=self.noise_generator, device=dev)
                    # NB: the reference swallows the caller's CFG scales and always runs 3.0 / 1.1
                    out = generate_image(self.model, self.rf, self.vision, x, cache_len, torch.cat((am, one), 1), unc, tunc,
                                         noises, temperature=image_gen_temperature, text_cfg=3.0, image_cfg=1.1)
                    cache_len = out["cache_len"]
                    hidden = out["last_hidden"][0:1]
                    pil = tensor_to_pil(out["image"])
                    for i in range(100):                                 # modeling_bailing_moe.py:1788-1796
                        name = f"{output_image_prefix}.png" if i == 0 else f"{output_image_prefix}_{i}.png"
                        if not os.path.exists(name):
                            print(f"Saving to {name}")
                            pil.save(name)
                            break
                    n_img += 1
                    self.last_image = out["image"]
                    self.last_generation = out                           # latents / sem / last_hidden of the image (inspection, tests)
                    break                                                # speculative tokens behind `<image>` are dropped
            else:
                cache_len += n                                           # every token of the chunk was fed
        # state for the next round (:273-299)
        self.past_len = cache_len
        pad1 = torch.ones(1, cache_len - prompt_mask_len, dtype=attention_mask.dtype)
        pad0 = torch.zeros(1, cache_len - prompt_mask_len, dtype=attention_mask.dtype)
        if os.environ.get("PAST_MODE", "DROP") == "KEEP":
            self.past_attention_mask = torch.cat((attention_mask, pad1), dim=1)
            self.past_text_uncond_attention_mask = torch.cat((tunc, pad1), dim=1)
            self.past_uncond_attention_mask = torch.cat((unc, pad0), dim=1)
        else:
            self.past_attention_mask = torch.cat((attention_mask, pad1), dim=1)
            self.past_text_uncond_attention_mask = torch.cat((attention_mask, pad1), dim=1)
            self.past_uncond_attention_mask = torch.cat((attention_mask, pad0), dim=1)
        return torch.cat((input_ids.cpu(), torch.tensor([new_ids], dtype=input_ids.dtype)), dim=1)
