"""Shared helpers for the test-suite: golden fixtures + synthetic state dicts."""
import json
import os

import numpy as np
import torch

from ming_univision_amd import configuration as C
from ming_univision_amd.synth import synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in "US" and v.ndim == 0:
            out[k] = json.loads(str(v))
        elif v.ndim == 0:
            out[k] = v.item()
        else:
            out[k] = torch.from_numpy(v)
    return out


def checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def mingtok_sd(cfg_dict, seed, device="cpu", dtype=torch.float32):
    return synth_state_dict(C.mingtok_param_shapes(C.MingTokConfig(**cfg_dict)), seed, device, dtype)


def llm_sd(llm_dict, rf_dict, seed, device="cpu", dtype=torch.float32):
    return synth_state_dict(C.llm_param_shapes(C.BailingMoeConfig(**llm_dict), rf_dict, 32), seed, device, dtype)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def row_errs(a, b):
    """Per-row relative error: max|a_r - b_r| / max|b_r| over the last dimension's rows (every leading index is a row).  What
    `rel_err`'s global max-norm cannot see: a small-magnitude row that is far off."""
    a, b = a.double().cpu(), b.double().cpu()
    a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    return (a - b).abs().amax(1) / (b.abs().amax(1) + 1e-30)


def rel_err_rows(a, b):
    """The worst row of `row_errs`."""
    return float(row_errs(a, b).max())


class DecoderBackedSD(dict):
    """The oracle's view of a BailingMoeDecoder that lives on the GPU: a state dict under the reference's parameter names
    (`model.layers.{i}.…`, modeling_bailing_moe.py) whose per-layer entries are pulled from the decoder's packed bf16 device tensors
    on first use (the inverse of `pack_experts`) as fp32 CPU tensors and dropped when another layer is asked for — the 16B-A3B stack
    is 32 GB of bf16, the oracle walks it layer by layer.  Entries stored with `__setitem__` (RF head, vis_head, final norm) stay.
    Test infrastructure."""

    def __init__(self, dec, extra=None):
        super().__init__()
        self.dec, self._layer, self._held = dec, None, {}
        self["model.norm.weight"] = dec.final_norm.float().cpu()
        if dec.word_embeddings is not None:
            self["model.word_embeddings.weight"] = dec.word_embeddings.float().cpu()
        if dec.lm_head is not None:
            self["lm_head.weight"] = dec.lm_head.float().cpu()
        for k, v in (extra or {}).items():
            self[k] = v

    def __missing__(self, name):
        parts = name.split(".")
        assert parts[0] == "model" and parts[1] == "layers", name
        li = int(parts[2])
        if li != self._layer:
            self._layer, self._held = li, {}
        if name not in self._held:
            self._held[name] = self._pull(li, ".".join(parts[3:])).cpu().float()
        return self._held[name]

    def _pull(self, li, tail):
        ly, cfg = self.dec.layers[li], self.dec.cfg
        E, S, I = cfg.num_experts, self.dec.n_shared, cfg.moe_intermediate_size
        plain = {"input_layernorm.weight": "ln1", "attention.query_key_value.weight": "wqkv", "attention.dense.weight": "wdense",
                 "post_attention_layernorm.weight": "ln2", "mlp.gate.weight": "gate", "mlp.image_gate.weight": "image_gate"}
        if tail in plain:
            return ly[plain[tail]]
        gu, dn = ly["w_gate_up"], ly["w_down"]
        assert gu.dtype == torch.bfloat16, "bf16 decoders only (8-bit ones: dequantized_state_dict)"
        t = tail.split(".")
        if t[1] == "experts":
            e, which = int(t[2]), t[3]
            return {"gate_proj": gu[e, :I], "up_proj": gu[e, I:], "down_proj": dn[e]}[which]
        assert t[1] == "shared_experts", tail
        which = t[2]
        if which == "down_proj":
            return torch.cat([dn[E + s] for s in range(S)], 1)
        return torch.cat([gu[E + s, :I] if which == "gate_proj" else gu[E + s, I:] for s in range(S)], 0)


class OracleConversation:
    """The reference's multi-round flow (MingUniVisionForConditionalGeneration.generate, modeling_bailingmm.py:206-301, over
    BailingMoeForCausalLM.forward's `<image>` branch, modeling_bailing_moe.py:1769-1796) driven with the oracle's pieces: one KV
    cache and three attention masks carried from round to round (PAST_MODE KEEP / DROP, :273-299), greedy text tokens, an image
    whenever `<image>` is emitted (or forced as the first token).  Test infrastructure.  PINNED (round 6) to the reference's own
    generate on a three-round conversation under KEEP and DROP: tests/golden/multiround_tiny.npz (oracle/gen_golden.gen_multiround),
    tests/test_oracle_golden.py::test_multiround_state_machine_vs_reference."""

    def __init__(self, sd, lsd, tsd, ocfg, steps, past_mode="DROP", eos_token_id=None, decode_pixels=False):
        from oracle import bailing_ref, mingtok_ref
        self.decode_pixels = decode_pixels
        self.B, self.M = bailing_ref, mingtok_ref
        self.sd, self.lsd, self.tsd, self.cfg, self.steps, self.mode, self.eos = sd, lsd, tsd, ocfg, steps, past_mode, eos_token_id
        self.kvs = bailing_ref.new_kv(ocfg)
        self.past = None                     # (am, unc, tunc) of the rounds so far
        self.cache_len = 0

    def _emb(self, ids):
        return self.sd["model.word_embeddings.weight"][ids]

    def round(self, ids, unc=None, tunc=None, pixel_values=None, patch_id=None, max_new_tokens=2, forced_first_token=None, noises=None):
        """ids [1, T]; unc / tunc [1, T] (default: ones).  Returns dict(tokens, images=[generate_image results], prompt_hidden)."""
        B, cfg = self.B, self.cfg
        T = ids.shape[1]
        am = torch.ones(1, T, dtype=torch.long)
        unc = am.clone() if unc is None else unc
        tunc = am.clone() if tunc is None else tunc
        if self.past is not None:
            am, unc, tunc = (torch.cat((p, m), 1) for p, m in zip(self.past, (am, unc, tunc)))
        prompt_mask_len = am.shape[1]
        emb = self._emb(ids).clone()
        image_mask = None
        if pixel_values is not None:
            feat = self.M.mingtok_forward(pixel_values, self.tsd)["x_norm_patchtokens"]
            image_mask = ids == patch_id
            emb[image_mask] = B.linear_proj(feat.float(), self.lsd).reshape(-1, cfg.hidden_size)
        h = B.model_forward(emb, self.sd, cfg, None, None, self.kvs, image_mask=image_mask)[:, -1:]
        prompt_hidden = h[:, 0]
        cache_len = self.cache_len + T
        toks, images = [], []
        one = torch.ones(1, 1, dtype=torch.long)
        while len(toks) < max_new_tokens:
            tok = int(B.lm_logits(h, self.sd).argmax())
            if not toks and forced_first_token is not None:
                tok = int(forced_first_token)
            toks.append(tok)
            if tok == self.eos or len(toks) == max_new_tokens:
                break                                                   # the call's last token is never fed
            if tok == cfg.image_start_token:
                if am.shape[1] < cache_len:
                    am = torch.cat((am, torch.ones(1, cache_len - am.shape[1], dtype=torch.long)), 1)
                caches = self.M.semdec_new_cache(self.tsd)
                out = B.generate_image(self._emb(torch.tensor([[tok]])), self.kvs, torch.cat((am, one), 1), unc, tunc, self.sd, cfg,
                                       noises[len(images)],
                                       latent_to_sem=lambda lat: self.M.mingtok_feature_decoder_step(lat, self.tsd, caches),
                                       linear_proj=lambda s_: B.linear_proj(s_, self.lsd),
                                       sem_to_pix=(lambda s_: self.M.pixel_decoder_forward(s_, self.tsd)) if self.decode_pixels else (lambda s_: None),
                                       steps=self.steps)
                images.append(out)
                cache_len += 1 + cfg.num_image_tokens_for_gen
                h = out["last_hidden"][0:1, -1:]
                continue
            h = B.model_forward(self._emb(torch.tensor([[tok]])), self.sd, cfg, None, None, self.kvs)[:, -1:]
            cache_len += 1
        self.cache_len = cache_len
        am0, unc0, tunc0 = am[:, :prompt_mask_len], unc, tunc
        pad1 = torch.ones(1, cache_len - prompt_mask_len, dtype=torch.long)
        pad0 = torch.zeros(1, cache_len - prompt_mask_len, dtype=torch.long)
        if self.mode == "KEEP":
            self.past = (torch.cat((am0, pad1), 1), torch.cat((unc0, pad0), 1), torch.cat((tunc0, pad1), 1))
        else:
            self.past = (torch.cat((am0, pad1), 1), torch.cat((am0, pad0), 1), torch.cat((am0, pad1), 1))
        return dict(tokens=toks, images=images, prompt_hidden=prompt_hidden)
