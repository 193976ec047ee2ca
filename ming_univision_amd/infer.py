"""MingUniVisionInfer — the inference façade (mingunivision/mingunivisioninfer.py:28-119).

    infer = MingUniVisionInfer(model_name_or_path)            # or MingUniVisionInfer(None) for synthetic weights
    text = infer.generate(messages, max_new_tokens=512, output_image_prefix="output", for_edit=False)
    infer.reset_inner_state()

`model_name_or_path` is a directory with `config.json` (MingUniVisionConfig incl. `vishead_diffloss_config`)
and `*.safetensors` shards keyed by the reference's parameter names; the tokenizer is read from the same
directory when a `tokenizer.json` is present, otherwise the byte-level stand-in is used.

`dtype` is the reference's weight-format switch (:46-70: "bf16", or weight-only "int8" / "int4" through quanto / bitsandbytes):
  "bf16" (default);
  "int4" — bitsandbytes NF4 restated: every nn.Linear weight (lm_head included — the reference passes its own
           skip list, which replaces HF's default one) becomes bf16(NF4[code] * absmax) with one fp32 absmax per 64 weights.  The RF
           head's ResBlock / adaLN matrices and the decoder's experts — 95 % of the bytes a visual token streams — live in HBM as 4-bit
           codes and are decoded inside the weight-streaming kernels; the other Linears hold the same model's values as bf16;
  "int8" — optimum-quanto's qint8 rule restated: every nn.Linear weight becomes bf16(s * q) with
           a bf16-valued scale s = amax / 127 per output row and q = clamp(round(W / s), -128, 127); experts, RF ResBlock and adaLN
           matrices are streamed as int8 bytes and decoded in the kernels, the other Linears hold the same values as bf16;
  "fp8"  — the same tensors as OCP e4m3 bytes + power-of-two row scales (no reference counterpart).
All modes quantise once at load from the same bf16 checkpoint; the arithmetic (fp32 / bf16 hi+lo activations, bf16 MFMA, fp32
accumulate) is unchanged.
"""
import glob
import os

import torch

from .configuration import MingUniVisionConfig
from .modeling import MingUniVisionForConditionalGeneration
from .processing import BailingMMProcessor, SpecialTokenTokenizer


def load_safetensors_dir(path):
    from safetensors import safe_open
    sd = {}
    for fn in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
        with safe_open(fn, framework="pt", device="cpu") as f:
            for k in f.keys():
                sd[k] = f.get_tensor(k)
    return sd


class HFTokenizerAdapter:
    """Thin adapter over `tokenizers.Tokenizer` exposing what BailingMMProcessor needs."""

    def __init__(self, path):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(path)
        self.chat_template = None

    def convert_tokens_to_ids(self, tok):
        return self.tk.token_to_id(tok)

    def encode(self, text, add_special_tokens=False):
        return self.tk.encode(text, add_special_tokens=add_special_tokens).ids

    def __call__(self, text, **kw):
        if isinstance(text, str):
            text = [text]
        ids = [self.encode(t) for t in text]
        return {"input_ids": ids, "attention_mask": [[1] * len(i) for i in ids]}

    def decode(self, ids, skip_special_tokens=True, **kw):
        return self.tk.decode([int(i) for i in ids], skip_special_tokens=skip_special_tokens)

    def batch_decode(self, seqs, **kw):
        return [self.decode(s, **kw) for s in seqs]


class MingUniVisionInfer:
    def __init__(self, model_name_or_path=None, dtype="bf16", device="cuda", config=None, seed=0, t_max=4096):
        if dtype not in ("bf16", "fp8", "int8", "int4"):
            raise ValueError(f"dtype={dtype!r}: 'bf16', 'int4' (bitsandbytes NF4), 'int8' or 'fp8' (weight-only byte formats)")
        self.model_name_or_path = model_name_or_path
        self.dtype = dtype
        self.model, self.tokenizer, self.processor = self.load_model_processor(config, device, seed, t_max)
        self.model.tokenizer = self.tokenizer

    def load_model_processor(self, config, device, seed, t_max):
        path = self.model_name_or_path
        tokenizer = None
        sd = None
        if path is not None and os.path.isdir(path):
            if config is None:
                config = MingUniVisionConfig.from_pretrained(path)
            tj = os.path.join(path, "tokenizer.json")
            if os.path.exists(tj):
                tokenizer = HFTokenizerAdapter(tj)
            if glob.glob(os.path.join(path, "*.safetensors")):
                sd = load_safetensors_dir(path)
        if config is None:
            config = MingUniVisionConfig.ming_univision_16b_a3b()
        if tokenizer is None:
            tokenizer = SpecialTokenTokenizer()
        processor = BailingMMProcessor(tokenizer=tokenizer)
        model = MingUniVisionForConditionalGeneration(config, state_dict=sd, device=device, seed=seed, t_max=t_max, weights=self.dtype)
        return model, tokenizer, processor

    def generate(self, messages, max_new_tokens=512, output_image_prefix="output", for_edit=False, **kw):
        text = self.processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, use_system=True)
        image_inputs, _, _ = self.processor.process_vision_info(messages)
        inputs = self.processor(text=[text], images=image_inputs, return_tensors="pt",
                                image_patch_size=self.model.vision.patch_size, for_edit=for_edit)
        with torch.no_grad():
            generated_ids = self.model.generate(**inputs, max_new_tokens=max_new_tokens, use_cache=True,
                                                output_image_prefix=output_image_prefix, **kw)
        trimmed = [out_ids[len(in_ids):] for in_ids, out_ids in zip(inputs["input_ids"], generated_ids)]
        return self.processor.batch_decode(trimmed, skip_special_tokens=True, clean_up_tokenization_spaces=False)[0]

    def generate_batch(self, messages_list, output_image_prefixes=None, **kw):
        """Extension: B independent text->image conversations generated in lock-step (one pass of every weight serves all of
        them; > 32 images take the wide MFMA route).  Same chat template / processor as `generate`; returns the PNG names."""
        reqs = []
        for messages in messages_list:
            text = self.processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, use_system=True)
            reqs.append(self.processor(text=[text], images=None, return_tensors="pt", image_patch_size=self.model.vision.patch_size,
                                       for_edit=False))
        with torch.no_grad():
            return self.model.generate_image_batch(reqs, output_image_prefixes=output_image_prefixes, **kw)["files"]

    def generate_batch_text(self, messages_list, max_new_tokens=64, **kw):
        """Extension: B independent understanding / chat conversations (text and images in, text out) decoded in lock-step.
        Returns the list of decoded answers."""
        reqs = []
        for messages in messages_list:
            text = self.processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, use_system=True)
            image_inputs, _, _ = self.processor.process_vision_info(messages)
            reqs.append(self.processor(text=[text], images=image_inputs, return_tensors="pt",
                                       image_patch_size=self.model.vision.patch_size, for_edit=False))
        with torch.no_grad():
            ids = self.model.generate_text_batch(reqs, max_new_tokens=max_new_tokens, **kw)
        return self.processor.batch_decode(ids, skip_special_tokens=True, clean_up_tokenization_spaces=False)

    def reset_inner_state(self):
        self.model.reset_inner_state()
