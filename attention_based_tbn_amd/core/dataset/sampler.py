"""Segment-index sampling (host side, integer exact).

Mirror of the index-selection part of reference `Video_Dataset.__getitem__`
(core/dataset/dataset.py:155-172), `_get_offsets` (:194-239) and the per-modality frame
arithmetic of `EpicVideoRecord` (core/dataset/epic_record.py:25-46).  Train-mode offsets come from
NumPy's global legacy RandomState with one `randint` draw per call, in modality order, so a run
seeded like the reference's `main.py:20` reproduces its indices bit for bit.
"""
import numpy as np

_MODALITIES = ("RGB", "Flow", "Audio")


def frame_span(start_frame, stop_frame):
    """annotation (1-based start_frame, stop_frame) -> ({modality: first index}, {modality: #frames})"""
    first = {"RGB": start_frame - 1, "Flow": (start_frame - 1) // 2, "Audio": start_frame - 1}
    last = {"RGB": stop_frame - 2, "Flow": (stop_frame - 2) // 2, "Audio": stop_frame - 2}
    return first, {m: last[m] - first[m] for m in _MODALITIES}


def get_offsets(first, num_frames, modality, mode, num_segments, frame_len):
    span = num_frames - frame_len + 1 if mode == "train" else num_frames
    seg_len = span // num_segments
    if seg_len <= 0:
        return first + np.zeros((num_segments), dtype=np.int64)
    if mode == "train":
        off = np.random.randint(seg_len, size=num_segments)
    else:
        off = seg_len // 2
        if modality == "Flow":
            off = max(off - (frame_len // 2), 0)   # centre the stacked-flow window, never negative
    return (first + np.arange(0, num_segments) * seg_len + off).astype(np.int64)


class SegmentSampler:
    """cfg-driven wrapper: `sampler(start_frame, stop_frame) -> OrderedDict-like {modality: int64[n]}`"""

    def __init__(self, cfg, modality, mode="train"):
        self.modality = list(modality)
        self.mode = mode
        self.sampling = cfg.data.sampling
        self.num_segments = {"train": cfg.train.num_segments, "val": cfg.val.num_segments,
                             "test": cfg.test.num_segments}[mode]
        self.frame_len = {m: (cfg.data.flow.win_length if m == "Flow" else 1) for m in self.modality}

    def __call__(self, start_frame, stop_frame):
        first, num = frame_span(start_frame, stop_frame)
        out = {}
        for i, m in enumerate(self.modality):
            if i > 0 and self.sampling == "sync":
                idx = out[self.modality[0]]
                out[m] = (idx / 2).astype(np.int64) if m == "Flow" else idx
            else:
                out[m] = get_offsets(first[m], num[m], m, self.mode, self.num_segments, self.frame_len[m])
        return out

    def flow_frames(self, indices):
        """expand flow start indices into the stacked window (dataset.py:168-172)"""
        L = self.frame_len["Flow"]
        return (indices.repeat(L) + np.tile(np.arange(L), self.num_segments)).astype(np.int64)
