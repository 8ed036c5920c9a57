from .sampler import SegmentSampler, frame_span, get_offsets  # noqa: F401
from .spectrogram import Spectrogram, trim_audio, trim_audio_window  # noqa: F401
from .prior import attention_prior, gaussian_kernel  # noqa: F401
from .transform import (CenterCrop, DevicePipeline, MultiScaleCrop, RandomCrop, RandomHorizontalFlip,  # noqa: F401
                        Rescale, get_transforms)
