"""Import shim for the *reference* Python sources under /root/reference.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py (in the build container,
where /root/reference exists) to run the reference's own PyTorch code on CPU and
record golden input/output vectors under tests/golden/.  Nothing here is
imported by the product package, by `-m gpu` tests, by smoke() or by bench.py:
/root/reference does not exist on the GPU box.

The reference pins transformers 4.52.4 / torch 2.7 and imports packages that are
absent from this image (omegaconf, torchvision, flash_attn).  The shim provides
the minimum so that the reference modules import and run unmodified:

  * stub modules `omegaconf` and `torchvision(.transforms(.functional))`
    (imported at module scope by mingtok/modeling_mingtok.py:2,
    mingtok/utils/processor.py:2-5, mingunivision/modeling_bailing_moe.py:33)
  * `transformers.utils.import_utils.is_torch_fx_available` (removed in 5.x,
    used at modeling_bailing_moe.py:61)
  * `LegacyDynamicCache`: the 4.52 `DynamicCache` surface the reference uses
    (key_cache / value_cache / update / get_seq_length / get_usable_length,
    modeling_bailing_moe.py:778,789,1896-1902,1993; attention.py:150)
  * round 6, for the multi-round state machine (`install_multimodal`, `cache_position_452`): import-only stubs of the audio
    packages modeling_bailingmm.py:22 / modeling_utils.py:17 pull in (funasr, whisper — no arithmetic, never called on the
    vision path) and the ONE piece of the transformers-4.52.4 `generate` contract that 5.x dropped and the reference's
    prepare_inputs_for_generation (modeling_bailing_moe.py:1968-2080) relies on: `cache_position` kept in model_kwargs
    (4.52.4 GenerationMixin._get_initial_cache_position / _update_model_kwargs_for_generation).  With it the installed
    transformers' own greedy loop drives the reference's generate / forward / prepare_inputs_for_generation UNMODIFIED.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MING_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "mingtok"))


def _stub_module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Make `import mingtok...` and `import modeling_bailing_moe` work."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    import torch  # noqa: F401
    import transformers  # noqa: F401  (must be imported before the stubs)
    import transformers.utils.import_utils as iu

    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    import transformers.utils as tu

    if not hasattr(tu, "is_torch_fx_available"):
        tu.is_torch_fx_available = lambda: False

    if "omegaconf" not in sys.modules:
        class _Missing:  # sentinel used as dataclass default
            pass

        class _OmegaConf:
            @staticmethod
            def create(x=None):
                return x

            @staticmethod
            def to_container(x, **kw):
                return x

            @staticmethod
            def load(path):
                raise RuntimeError("omegaconf stub: load unsupported")

        _stub_module("omegaconf", MISSING=_Missing(), OmegaConf=_OmegaConf,
                     DictConfig=dict, ListConfig=list)
    if "torchvision" not in sys.modules:
        tv = _stub_module("torchvision")
        tvt = _stub_module("torchvision.transforms")
        tvf = _stub_module("torchvision.transforms.functional")

        class _InterpolationMode:
            BICUBIC = "bicubic"
            BILINEAR = "bilinear"

        tvt.InterpolationMode = _InterpolationMode
        tvf.InterpolationMode = _InterpolationMode
        for nm in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "ToPILImage"):
            setattr(tvt, nm, type(nm, (), {"__init__": lambda self, *a, **k: None}))
        tv.transforms = tvt
        tvt.functional = tvf
        tv.io = _stub_module("torchvision.io")
        tv.__version__ = "0.0-stub"
    if "torchaudio" not in sys.modules:
        _stub_module("torchaudio")

    for p in (REFERENCE_ROOT, os.path.join(REFERENCE_ROOT, "mingunivision")):
        if p not in sys.path:
            sys.path.insert(0, p)
    _installed = True


def make_legacy_cache():
    """A DynamicCache with the transformers-4.52 attribute surface."""
    import torch
    from transformers.cache_utils import Cache

    class LegacyDynamicCache(Cache):
        def __init__(self):
            # do not call Cache.__init__ (5.x wants layer classes)
            self.key_cache = []
            self.value_cache = []
            self._seen_tokens = 0

        @property
        def seen_tokens(self):
            return self._seen_tokens

        def __len__(self):
            return len(self.key_cache)

        def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
            if layer_idx == 0:
                self._seen_tokens += key_states.shape[-2]
            if len(self.key_cache) <= layer_idx:
                self.key_cache.append(key_states)
                self.value_cache.append(value_states)
            else:
                self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], key_states], dim=-2)
                self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], value_states], dim=-2)
            return self.key_cache[layer_idx], self.value_cache[layer_idx]

        def get_seq_length(self, layer_idx=0):
            if len(self.key_cache) <= layer_idx:
                return 0
            return self.key_cache[layer_idx].shape[-2]

        def get_max_length(self):
            return None

        def get_max_cache_shape(self):
            return None

        def get_usable_length(self, new_seq_length, layer_idx=0):
            return self.get_seq_length(layer_idx)

        def to_legacy_cache(self):
            return tuple((k, v) for k, v in zip(self.key_cache, self.value_cache))

        # what transformers 5.x's generate asks a cache object before it starts (no arithmetic)
        is_compileable = False

        @property
        def layers(self):
            return [None] * len(self.key_cache)

    return LegacyDynamicCache()


def install_multimodal():
    """Make `import modeling_bailingmm` work: import-only stubs for the audio towers it pulls in at module scope
    (modeling_bailingmm.py:22 `funasr.models.sanm.encoder.SANMEncoder`, modeling_utils.py:17 `whisper.model.AudioEncoder`).
    Neither class is instantiated or called on the vision / text path."""
    install()
    import torch.nn as nn
    for n in ("funasr", "funasr.models", "funasr.models.sanm"):
        if n not in sys.modules:
            _stub_module(n)
    if "funasr.models.sanm.encoder" not in sys.modules:
        _stub_module("funasr.models.sanm.encoder", SANMEncoder=type("SANMEncoder", (nn.Module,), {}))
    if "whisper" not in sys.modules:
        _stub_module("whisper")
        _stub_module("whisper.model", AudioEncoder=type("AudioEncoder", (nn.Module,), {}))


def cache_position_452(lm, trace=None):
    """Wrap `lm.prepare_inputs_for_generation` (the REFERENCE's function, untouched) so that it receives what transformers 4.52.4's
    generate would hand it and 5.x no longer does:
      * `cache_position`: first call of a generate = arange(len(inputs_embeds or input_ids))[cache.get_seq_length():]
        (GenerationMixin._get_initial_cache_position), every later call = previous[-1:] + 1 (_update_model_kwargs_for_generation;
        an EMPTY first value stays empty — that is what a follow-up round on a longer cache gets, and the reference's slicing
        rules :2001-2016 are written around it);
      * `position_ids` = None as the reference's caller passes it (modeling_bailingmm.py:254; 5.x keeps a running tensor there);
      * 5.x-only kwargs (`next_sequence_length`, `is_first_iteration`) are dropped.
    trace (a list): appended per call with (len(input_ids seen), cache length before, attention-mask length out, n new ids,
    first position id) — the bookkeeping the fixture pins.  Returns `start()`: call it before every generate."""
    import functools

    import torch
    orig = lm.prepare_inputs_for_generation
    state = {"cp": None}

    @functools.wraps(orig)
    def prep(input_ids, **kw):
        kw.pop("next_sequence_length", None)
        kw.pop("is_first_iteration", None)
        kw["position_ids"] = None
        if state["cp"] is None:
            emb = kw.get("inputs_embeds")
            n = emb.shape[1] if emb is not None else input_ids.shape[1]
            cp = torch.ones(n, dtype=torch.int64).cumsum(0) - 1
            pkv = kw.get("past_key_values")
            if pkv is not None and pkv.get_seq_length() is not None:
                cp = cp[pkv.get_seq_length():]
            state["cp"] = cp
        else:
            state["cp"] = state["cp"][-1:] + 1
        before = kw["past_key_values"].get_seq_length()
        out = orig(input_ids, cache_position=state["cp"], **kw)
        if trace is not None:
            n_new = out["input_ids"].shape[1] if out.get("input_ids") is not None else out["inputs_embeds"].shape[1]
            trace.append((int(input_ids.shape[1]), int(before), int(out["attention_mask"].shape[1]), int(n_new),
                          int(out["position_ids"][0, 0])))
        return out
    lm.prepare_inputs_for_generation = prep

    def start():
        state["cp"] = None
    return start
