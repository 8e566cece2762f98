"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU (torch fp32 / fp64, numpy) restatement of the reference algorithm for the CGG hot path
(jianzongwu/betrayed-by-captions, open_set/models + the mmcv 1.7.1 / mmdet 2.28.2 leaf ops it
selects by `type=` string). It exists only so that tests can check the HIP path against it.

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` -- as the checker / the timed CPU baseline, never as the thing shipped. Nothing under
`betrayed-by-captions_amd/` imports it, and the product path raises when libcgg_hip.so is missing.

Parity status (see DESIGN.md "Oracle"):
  * Tier A -- reference files that execute verbatim in the build container
    (losses/grounding_loss.py, transformers/transformers.py, transformers/caption_tranformer.py,
    utils/bert_embeddings.py and, through the import shim of tests/golden/make_golden.py,
    mask2former_head.py, maskformer_fusion_head.py, assigners/mask_hungarian_assigner.py):
    the oracle is PINNED against golden vectors generated from them (tests/golden/*.npz).
  * Tier B -- mmcv / mmdet leaf ops (MSDeformAttn, MSDeformAttnPixelDecoder, transformer layers,
    match costs, losses): their source is NOT under /root/reference and the reference ships no
    tests, so for them "parity unpinned" by the reference; they are pinned instead against PyTorch
    primitives (F.grid_sample, nn.MultiheadAttention, F.interpolate, scipy linear_sum_assignment)
    and an independent scalar-loop implementation.
"""
