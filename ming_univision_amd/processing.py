"""Host-side pre-processing, mirroring mingunivision/processing_bailingmm.py.

Kept surface: `BailingMMProcessor.__call__(images, text, for_edit, image_patch_size)`,
`.apply_chat_template`, `.process_vision_info`, `.tokenize`, `.batch_decode`.
Host integer/string logic only (no GPU work): chat packing, `<IMAGE>` expansion into
`<image>` + N x `<imagePatch>` + `</image>`, the 1024^2 "understand" vs 512^2 "gen/edit" image
transforms and the classifier-free-guidance attention masks (uncond / text-uncond).

The reference tokenizer blob (`tokenizer.json`) is not available offline, so the tokenizer is
pluggable: anything exposing `encode(text, add_special_tokens=False) -> list[int]`,
`convert_tokens_to_ids(tok)`, `decode(ids, skip_special_tokens)`; `SpecialTokenTokenizer` is a
self-contained stand-in that knows the reference's special-token ids (tokenizer_config.json) and
byte-encodes everything else.
"""
import re

import numpy as np
import torch

DEFAULT_IMAGE_PATCH_TOKEN = "<imagePatch>"
DEFAULT_IM_START_TOKEN = "<image>"
DEFAULT_IM_END_TOKEN = "</image>"
USER_PREFIX = "<role>HUMAN</role>"
ASSISTANT_PREFIX = "<role>ASSISTANT</role>"

# ids from mingunivision/tokenizer_config.json / special_tokens_map.json
SPECIAL_TOKEN_IDS = {
    "<|endoftext|>": 126081, "<role>": 126340, "</role>": 126341,
    "<imagePatch>": 126346, "<image>": 126347, "</image>": 126348,
}


class SpecialTokenTokenizer:
    """Stand-in tokenizer: reference special tokens -> their ids, other text -> UTF-8 bytes (ids 0..255),
    "HUMAN"/"ASSISTANT" -> two reserved ids so that the role tags are fixed-length like the real BPE's."""
    WORDS = {"HUMAN": 300, "ASSISTANT": 301}

    def __init__(self, special=None):
        self.special = dict(special or SPECIAL_TOKEN_IDS)
        self.inv = {v: k for k, v in self.special.items()}
        self.inv.update({v: k for k, v in self.WORDS.items()})
        toks = sorted(list(self.special) + list(self.WORDS), key=len, reverse=True)
        self._re = re.compile("(" + "|".join(re.escape(t) for t in toks) + ")")
        self.chat_template = None

    def convert_tokens_to_ids(self, tok):
        return self.special[tok]

    def encode(self, text, add_special_tokens=False):
        ids = []
        for part in self._re.split(text):
            if not part:
                continue
            if part in self.special:
                ids.append(self.special[part])
            elif part in self.WORDS:
                ids.append(self.WORDS[part])
            else:
                ids.extend(part.encode("utf-8"))
        return ids

    def __call__(self, text, **kw):
        if isinstance(text, str):
            text = [text]
        ids = [self.encode(t) for t in text]
        return {"input_ids": ids, "attention_mask": [[1] * len(i) for i in ids]}

    def decode(self, ids, skip_special_tokens=True, **kw):
        out, buf = [], bytearray()
        for i in (int(x) for x in ids):
            if i < 256:
                buf.append(i)
                continue
            if buf:
                out.append(buf.decode("utf-8", errors="replace"))
                buf = bytearray()
            if not skip_special_tokens or i in self.WORDS.values():
                out.append(self.inv.get(i, f"<{i}>"))
        if buf:
            out.append(buf.decode("utf-8", errors="replace"))
        return "".join(out)

    def batch_decode(self, seqs, **kw):
        return [self.decode(s, **kw) for s in seqs]


def _pil_to_tensor(img, mean, std):
    a = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0            # ToTensor
    t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
    m = torch.tensor(mean).view(3, 1, 1)
    s = torch.tensor(std).view(3, 1, 1)
    return (t - m) / s                                                      # Normalize


class MingTokUndProcessor:
    """Resize((S,S), bicubic) -> ToTensor -> Normalize (processing_bailingmm.py:80-100)."""

    def __init__(self, image_size=1024, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        self.image_size, self.mean, self.std = image_size, mean, std

    def __call__(self, img):
        from PIL import Image
        return _pil_to_tensor(img.resize((self.image_size, self.image_size), Image.BICUBIC), self.mean, self.std)


class MingTokCenterCropProcessor:
    """Resize(S, bicubic: shorter side -> S) -> CenterCrop(S) -> ToTensor -> Normalize
    (processing_bailingmm.py:102-123; mingtok/utils/processor.py:8-30)."""

    def __init__(self, image_size=512, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        self.image_size, self.mean, self.std = image_size, mean, std

    def __call__(self, img):
        from PIL import Image
        S = self.image_size
        w, h = img.size
        if w <= h:
            nw, nh = S, int(S * h / w)
        else:
            nw, nh = int(S * w / h), S
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - S) / 2.0)), int(round((nh - S) / 2.0))
        return _pil_to_tensor(img.crop((left, top, left + S, top + S)), self.mean, self.std)


CenterCropProcessor = MingTokCenterCropProcessor


def find_all_subsequences(sequence, subsequence):
    """Start offsets of every occurrence of `subsequence` (processing_bailingmm.py:364-372)."""
    seq, sub = list(sequence), list(subsequence)
    if not sub:
        return []
    return [i for i in range(len(seq) - len(sub) + 1) if seq[i:i + len(sub)] == sub]


def last_user_turn(seq, user_tag, assistant_tag):
    """(start, end, answered): the tokens of the LAST user turn — everything after the final user tag, up to the assistant tag that
    follows it (answered) or to the end of the sequence.  None when the sequence has no user tag."""
    users = find_all_subsequences(seq, user_tag)
    if not users:
        return None
    u = users[-1]
    start = u + len(user_tag)
    following = [a for a in find_all_subsequences(seq, assistant_tag) if a >= u]
    if following:
        return start, max(start, following[0]), True
    return start, len(seq), False


def cfg_attention_masks(seq, user_prefix_ids, assistant_prefix_ids, image_token_ids):
    """The two classifier-free-guidance masks BailingMMProcessor.tokenize derives from ONE id sequence
    (processing_bailingmm.py:304-352), expressed on the last user turn:
      uncond       hides the whole turn — but only when an assistant tag closes it;
      text-uncond  hides the turn's text and keeps its image tokens (`<image>`, `<imagePatch>`, `</image>`), closed or not."""
    seq = list(seq)
    uncond, text_uncond = [1] * len(seq), [1] * len(seq)
    turn = last_user_turn(seq, list(user_prefix_ids), list(assistant_prefix_ids))
    if turn is not None:
        start, end, answered = turn
        if answered:
            uncond[start:end] = [0] * (end - start)
        text_uncond[start:end] = [int(t in image_token_ids) for t in seq[start:end]]
    return uncond, text_uncond


class BatchFeature(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        return BatchFeature({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.items()})


class BailingMMProcessor:
    def __init__(self, image_processor=None, tokenizer=None, chat_template=None, image_token="<image>", **kwargs):
        self.tokenizer = tokenizer if tokenizer is not None else SpecialTokenTokenizer()
        self.image_processor = image_processor
        self.image_token = image_token
        self.chat_template = chat_template
        self.vis_processor = MingTokUndProcessor(image_size=1024)
        self.gen_processor = MingTokCenterCropProcessor(image_size=512)
        self.gen_terminator = [self.tokenizer.convert_tokens_to_ids("<|endoftext|>")]

    # -- chat packing (processing_bailingmm.py:374-437) ------------------------------------------
    def apply_system_template(self, text):
        return USER_PREFIX

    @staticmethod
    def _message_body(message):
        """Text of one message: its text items in order; an image item contributes `<IMAGE>` placeholders for the images the
        message's own text does not already mark with a literal `<image>` (the reference counts those on str(content), :392)."""
        marked = str(message["content"]).count("<image>")
        pieces = []
        for item in message["content"]:
            kind = item["type"]
            if kind == "text":
                pieces.append(item["text"])
            elif kind == "image":
                n = len(item["image"]) if isinstance(item["image"], (list, tuple)) else 1
                if n > marked:
                    pieces.append("\n".join(["<IMAGE>"] * (n - marked)))
            elif kind in ("video", "audio"):
                raise NotImplementedError("video/audio inputs are outside the MingTok hot path")
        return "".join(pieces)

    def apply_chat_template(self, conversation, system_template=None, **kwargs):
        """GLM-style packing (processing_bailingmm.py:377-437): the system template (= the first HUMAN tag) opens the text, a HUMAN
        turn is its bare body, an ASSISTANT turn is tag + body + `<|endoftext|>` + the next HUMAN tag; the generation prompt is a
        trailing ASSISTANT tag."""
        turns = []
        for message in conversation:
            role = message["role"]
            assert role in ("HUMAN", "ASSISTANT")
            body = self._message_body(message)
            turns.append(ASSISTANT_PREFIX + body + "<|endoftext|>" + USER_PREFIX if role == "ASSISTANT" else body)
        if kwargs.get("add_generation_prompt", True):
            turns.append(ASSISTANT_PREFIX)
        text = "".join(turns)
        head = system_template if system_template is not None else self.apply_system_template(text)
        return head + text

    def process_vision_info(self, conversations):
        """Collect PIL images of a conversation (image part of bailingmm_utils.process_vision_info:503-539)."""
        from PIL import Image
        images = []
        for message in conversations:
            if isinstance(message.get("content"), list):
                for ele in message["content"]:
                    if ele.get("type") == "image":
                        im = ele["image"]
                        for one in (im if isinstance(im, (list, tuple)) else [im]):
                            images.append(Image.open(one).convert("RGB") if isinstance(one, str) else one)
        return (images or None), None, None

    # -- <IMAGE> expansion (processing_bailingmm.py:445-464) ------------------------------------
    def _expand_image_tokens(self, text, image_grid_thw, special_token="<IMAGE>"):
        """Every `<IMAGE>` placeholder, in order over all samples, becomes `<image>` + one `<imagePatch>` per patch of the
        corresponding grid (t * h * w) + `</image>` + newline."""
        patches = iter(int(n) for n in torch.as_tensor(image_grid_thw).prod(dim=1))

        def block():
            return DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_PATCH_TOKEN * next(patches) + DEFAULT_IM_END_TOKEN + "\n"
        out = []
        for sample in text:
            head, *rest = sample.split(special_token)
            out.append(head + "".join(block() + part for part in rest))
        return out

    # -- tokenise + CFG masks (processing_bailingmm.py:282-361) ---------------------------------
    def tokenize(self, text, **kw):
        enc = self.tokenizer(text)
        input_ids, attention_mask = enc["input_ids"], enc["attention_mask"]
        if isinstance(input_ids, (list, tuple)) and not isinstance(input_ids[0], (list, tuple)):
            input_ids, attention_mask = [input_ids], [attention_mask]
        user_prefix_ids = self.tokenizer.encode(USER_PREFIX, add_special_tokens=False)
        assistant_prefix_ids = self.tokenizer.encode(ASSISTANT_PREFIX, add_special_tokens=False)
        image_token_ids = {self.tokenizer.convert_tokens_to_ids(t)
                           for t in (DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IM_END_TOKEN)}
        unc, tunc = [], []
        for seq in input_ids:
            m, tm = cfg_attention_masks(seq, user_prefix_ids, assistant_prefix_ids, image_token_ids)
            unc.append(m)
            tunc.append(tm)
        return {"input_ids": torch.tensor(input_ids, dtype=torch.long),
                "attention_mask": torch.tensor(attention_mask, dtype=torch.long),
                "uncond_attention_mask": torch.tensor(unc, dtype=torch.long),
                "text_uncond_attention_mask": torch.tensor(tunc, dtype=torch.long)}

    def __call__(self, images=None, videos=None, audios=None, text=None, for_edit=False, **kwargs):
        image_patch_size = kwargs.pop("image_patch_size", 32)
        if videos is not None or audios is not None:
            raise NotImplementedError("video/audio inputs are outside the MingTok hot path")
        if isinstance(text, str):
            text = [text]
        image_inputs = {}
        processor = self.gen_processor if for_edit else self.vis_processor
        if images is not None:
            tensors, grids = [], []
            for img in images:
                t = processor(img) if not isinstance(img, torch.Tensor) else img
                tensors.append(t)
                grids.append([1, t.shape[1] // image_patch_size, t.shape[2] // image_patch_size])
            image_inputs = {"pixel_values": torch.stack(tensors), "image_grid_thw": torch.tensor(grids)}
            text = self._expand_image_tokens(text, image_inputs["image_grid_thw"])
        return BatchFeature({**self.tokenize(text), **image_inputs})

    def batch_decode(self, *args, **kwargs):
        return self.tokenizer.batch_decode(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.tokenizer.decode(*args, **kwargs)
