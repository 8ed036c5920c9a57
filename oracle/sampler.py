"""Oracle (TEST INFRASTRUCTURE): segment-index sampling, NumPy, integer-exact.

Restates reference `core/dataset/epic_record.py:25-46` (frame arithmetic),
`core/dataset/dataset.py:194-239` (`_get_offsets`) and the sync/async rule and
flow window expansion of `core/dataset/dataset.py:156-172`.  Randomness is the
NumPy *global* legacy RandomState, one `randint` call per `_get_offsets` call
in train mode, exactly like the reference, so seeding `np.random.seed(s)` and
calling in the same order reproduces the reference indices bit for bit.
"""
import numpy as np


def frame_span(start_frame, stop_frame):
    """(start, num_frames) per modality from 1-based annotation columns (epic_record.py:25-46)."""
    start = {"RGB": start_frame - 1, "Flow": (start_frame - 1) // 2, "Audio": start_frame - 1}
    end = {"RGB": stop_frame - 2, "Flow": (stop_frame - 2) // 2, "Audio": stop_frame - 2}
    return start, {m: end[m] - start[m] for m in start}


def get_offsets(start, num_frames, modality, mode, num_segments, frame_len):
    """dataset.py:194-239.  `start`/`num_frames` are the per-modality scalars."""
    if mode == "train":
        seg_len = (num_frames - frame_len + 1) // num_segments
    else:
        seg_len = num_frames // num_segments
    if seg_len > 0:
        if mode == "train":
            offsets = np.random.randint(seg_len, size=num_segments)
        else:
            offsets = seg_len // 2
            if modality == "Flow":
                offsets = max(offsets - (frame_len // 2), 0)
        return (start + np.arange(0, num_segments) * seg_len + offsets).astype(np.int64)
    return start + np.zeros((num_segments), dtype=np.int64)


def sample_indices(start_frame, stop_frame, modalities, sampling, mode, num_segments, flow_win=5):
    """dataset.py:155-165: per-modality indices for one annotation row."""
    start, num = frame_span(start_frame, stop_frame)
    out = {}
    for i, m in enumerate(modalities):
        if i > 0 and sampling == "sync":
            out[m] = out[modalities[0]]
            if m == "Flow":
                out[m] = (out[m] / 2).astype(np.int64)
        else:
            fl = flow_win if m == "Flow" else 1
            out[m] = get_offsets(start[m], num[m], m, mode, num_segments, fl)
    return out


def flow_frame_indices(indices, flow_win, num_segments):
    """dataset.py:168-172: expand each flow index into `flow_win` consecutive frames."""
    return (indices.repeat(flow_win) + np.tile(np.arange(flow_win), num_segments)).astype(np.int64)
