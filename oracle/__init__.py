"""CPU oracle for the TBN hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / NumPy restatement of the reference
algorithm (tridivb/attention_based_tbn, `core/models/*`, the segment sampler
and the spectrogram of `core/dataset/dataset.py`).  It exists to *check* the
HIP product path; nothing in `attention_based_tbn_amd/` may import it.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it.

Parity pin: `tests/golden/make_golden.py` imports the unmodified reference
Python (with local stubs for cv2 / torchvision / pretrainedmodels / librosa)
in the build container and stores its outputs as fixtures under
`tests/golden/`; `tests/test_oracle_golden.py` checks this oracle against
those fixtures.  Two third-party pieces have no in-repo pin in the reference
(`pretrainedmodels.BNInception` stem/registration order and `librosa.stft`):
for those the oracle is the definition ("parity unpinned" at that boundary;
see DESIGN.md), cross-checked against the in-repo graph statement
`core/models/bn_inception_audio.py` from `conv2_3x3_reduce` onwards.
"""
