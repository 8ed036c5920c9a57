from .metric import Metric
from .misc import get_modality, get_time_diff, load_checkpoint, save_checkpoint, save_scores
from .optim import FusedSGD, clip_grad_norm_
from .train_step import TrainStep
from .warmup import GradualWarmupScheduler, build_optimizer

__all__ = ["Metric", "get_modality", "get_time_diff", "save_checkpoint", "load_checkpoint", "save_scores",
           "FusedSGD", "clip_grad_norm_", "TrainStep", "GradualWarmupScheduler",
           "build_optimizer"]
