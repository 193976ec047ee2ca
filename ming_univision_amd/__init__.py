"""ming_univision_amd — MI355X (gfx950) native hot path of Ming-UniVision.

MingTok-Vision tokenizer -> Bailing-MoE next-token forward -> rectified-flow SwiGLU head ->
MingTok decode, behind the reference's own Python surface.  All arithmetic runs in
libmingnative.so (hand-written HIP, C ABI in include/mingnative.h); PyTorch only owns device
memory and streams.  There is no CPU or eager fallback: importing an operator without the
built library raises.
"""
import os as _os

# generate_images(..., n_groups=G) runs G lock-step groups on G HIP streams; the ROCm runtime maps streams onto 4 hardware
# queues unless told otherwise (4 groups: 1447 vs 1686 tokens/s).  Only effective if HIP has not initialised yet.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .configuration import BailingMoeConfig, MingTokConfig, MingUniVisionConfig  # noqa: F401,E402

__all__ = ["BailingMoeConfig", "MingTokConfig", "MingUniVisionConfig", "MingTok", "MingUniVisionInfer",
           "MingUniVisionForConditionalGeneration", "BailingMMProcessor"]


def __getattr__(name):  # lazy: keep `import ming_univision_amd` cheap and torch-free
    if name == "MingTok":
        from .mingtok import MingTok
        return MingTok
    if name in ("MingUniVisionForConditionalGeneration",):
        from .modeling import MingUniVisionForConditionalGeneration
        return MingUniVisionForConditionalGeneration
    if name == "MingUniVisionInfer":
        from .infer import MingUniVisionInfer
        return MingUniVisionInfer
    if name == "BailingMMProcessor":
        from .processing import BailingMMProcessor
        return BailingMMProcessor
    raise AttributeError(name)
