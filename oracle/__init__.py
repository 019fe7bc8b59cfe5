"""CPU oracle for the Pair-Net inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pair-net_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported baseline.

Contents
--------
layers.py          fp32 torch-CPU restatement of the un-vendored third-party
                   layers the reference builds from its config (mmcv-full 1.7.0,
                   mmdet 2.25.1).  Unpinned against
                   those packages (not in /root/reference, not installed); the
                   arithmetic is pinned against HuggingFace transformers'
                   independent Mask2Former implementation (hf_pin.py).
hf_pin.py          state-dict converters oracle -> transformers for those pins.
swin.py            Swin backbone restatement, pinned to transformers.SwinBackbone.
matrix_learner.py  restatement of the reference's in-tree Matrix Learner
                   (pairnet/models/frameworks/cnn_factory.py:6-53); pinned
                   against the reference module imported by path.
head.py            restatement of CrossHead2.forward / get_bboxes
                   (pairnet/models/relation_heads/pairnet_head.py:216-417,
                   760-924); pinned against the reference class run under
                   name-only mmcv/mmdet shims (ref_shim.py).
ref_shim.py        loader that imports the reference's own pairnet_head.py from
                   /root/reference (this container only; never copied, never
                   shipped) with `mmcv`/`mmdet` replaced by stub modules whose
                   builders return the layers of layers.py.
make_golden.py     writes tests/golden/*.npz from the shimmed reference run.
"""
