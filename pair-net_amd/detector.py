"""Detector wrapper and result container: the caller side of the head.

`PSGTr` mirrors pairnet/models/frameworks/psgtr.py:73-88,148-156 (backbone -> head
-> `Result`), `triplet2Result` :15-51, `Result` the fields of
pairnet/models/relation_heads/approaches/relation_util.py:20-97 that the PSG
evaluator reads.  The backbone is SURVEY.md section 8 row a1 / (f)-2: the native
`ResNet50Hip` (backbone.py) or `SwinTransformerHip` (swin.py) by `backbone.type`.  There is
one implementation per component: the PyTorch-ROCm / MIOpen ResNet-50 that bench.py times
beside the native one lives in tools/torch_resnet50.py, not in the product package.
"""
from collections import OrderedDict

import contextlib
import queue
import threading
import weakref

import torch

from . import hip
from .config import ConfigDict
from .backbone import ResNet50Hip
from .baseline_head import CrossHeadBaseline
from .bbox_head import CrossHeadBBox
from .head import CrossHead2
from .neck import ChannelMapper
from .psgtr_head2 import PSGTrHead2
from .swin import SwinTransformerHip


class Result(object):
    """Scene-graph result of one image (numpy on the host, like the reference)."""

    FIELDS = ("refine_bboxes", "labels", "formatted_masks", "rel_pair_idxes", "rel_dists",
              "rel_labels", "pan_results", "masks", "bboxes", "dists", "rels",
              "refine_scores", "rel_scores", "triplet_scores", "img_shape", "sub_pos",
              "obj_pos")

    def __init__(self, **fields):
        for k in self.FIELDS:
            setattr(self, k, None)
        for k, v in fields.items():
            setattr(self, k, v)

    def is_none(self):
        return all(v is None for v in self.__dict__.values())

    # mmdet's result collection treats results as sequences (relation_util.py:87-97)
    def __len__(self):
        return 1

    def __getitem__(self, i):
        return self

    def __iter__(self):
        yield self


class _BoolArrayPool:
    """Host bool tensors handed out as numpy arrays nobody else holds, and taken back when
    such an array is garbage collected (`weakref.finalize`): a FRESH 49 MB host array per
    800x1333 image costs 6-8 ms of first-touch page faults (every fault / munmap of a process
    with a GPU context passes the amdgpu MMU notifier), a recycled one nothing.  A caller that
    keeps every result (mmdet's test loop does) never returns arrays, and each take allocates
    as `.cpu().numpy()` would."""

    def __init__(self, keep=3):
        self.keep, self.free = keep, {}

    def take(self, shape):
        free = self.free.get(tuple(shape))
        return free.pop() if free else torch.empty(tuple(shape), dtype=torch.bool)

    def _recycle(self, t):
        free = self.free.setdefault(tuple(t.shape), [])
        if len(free) < self.keep:
            free.append(t)

    def hand_out(self, t):
        arr = t.numpy()
        weakref.finalize(arr, self._recycle, t)     # back to the pool when `arr` dies
        return arr


class _MaskFetcher:
    """Device bool tensor -> numpy bool array for the synchronous `PSGTr.simple_test`, through
    the bit-packed transfer of `ResultStreamer` (pn_pack_bool_bits on the device, 1/8 of the
    bytes into a cached pinned buffer, pn_unpack_bits_host on host threads).

    The caller gets an array nobody else holds -- like the reference's `.cpu().numpy()` -- out
    of a `_BoolArrayPool`: 1.3 ms of expansion into a recycled array instead of 6-8 ms of page
    faults into a fresh one (tools/simple_test_probe.py)."""
    MIN = 1 << 20

    def __init__(self, threads=4, keep=3):
        self.threads = threads
        self.bits, self.pool = {}, _BoolArrayPool(keep)

    def __call__(self, t):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bool
                and t.numel() >= self.MIN):
            return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t
        nb = ((t.numel() + 7) // 8 + 15) // 16 * 16
        bkey = (t.device, nb)
        if bkey not in self.bits:
            if len(self.bits) >= 4:
                self.bits.clear()
            self.bits[bkey] = (torch.empty(nb, dtype=torch.uint8, device=t.device),
                               torch.empty(nb, dtype=torch.uint8, pin_memory=True))
        dev, host = self.bits[bkey]
        with torch.cuda.device(t.device):
            hip.pack_bool_bits(t.contiguous(), dev)
            host.copy_(dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        out = self.pool.take(t.shape)
        hip.unpack_bits_host(host, out, self.threads)
        return self.pool.hand_out(out)


def triplet2Result(triplets, use_mask, eval_mask_rels=False, mask_fetch=None):
    """8-tuple of `CrossHead2.get_bboxes` (or, without masks, the 6-tuple of
    `CrossHeadBBox.get_bboxes`) -> Result (psgtr.py:15-71).  `mask_fetch`: how the masks
    reach the host (default: `.cpu()`)."""
    np_ = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t   # (numpy passes)
    if not use_mask:
        bboxes, labels, rel_pairs, r_scores, r_labels, r_dists = triplets
        return Result(refine_bboxes=np_(bboxes), labels=np_(labels),
                      formatted_masks=dict(pan_results=None), rel_pair_idxes=np_(rel_pairs),
                      rel_dists=np_(r_dists), rel_labels=np_(r_labels), pan_results=None)
    bboxes, labels, rel_pairs, masks, pan_seg, r_scores, r_labels, r_dists = triplets
    pan_seg = np_(pan_seg)
    return Result(refine_bboxes=np_(bboxes), labels=np_(labels),
                  formatted_masks=dict(pan_results=pan_seg), rel_pair_idxes=np_(rel_pairs),
                  rel_dists=np_(r_dists), rel_labels=np_(r_labels), pan_results=pan_seg,
                  masks=(mask_fetch or np_)(masks))


class ResultStreamer:
    """`triplet2Result` (psgtr.py:15-51) for a pipeline of images.  The fields of a batch's
    `get_bboxes` tuples are staged device -> device into this ring's own buffers on the chain
    stream that produced them (which releases the pipeline slot at once) -- the 2R x H0 x W0
    bool masks, 49 of the 51 MB per 800x1333 image, as BITS (`pn_pack_bool_bits`) -- and copied
    to PINNED host buffers behind that on the same stream, under the next images' kernels; a
    worker thread of this object waits for the copy and expands the bits into the bool arrays
    the Results hold (`pn_unpack_bits_host`).  The host never allocates (a multi-MB host
    allocation per image is an mmap / munmap pair, and every munmap runs the amdgpu MMU
    notifier against the busy GPU: ~75 ms stalls measured, LABNOTES.md 6b).

        streamer = ResultStreamer(head, ring=4)
        streamer.push(results, pipe)      # results of PipelinedHead.submit() / get_bboxes()
        ...
        for r in streamer.pop():          # oldest pushed batch -> [Result], same fields and
            ...                           # dtypes as triplet2Result

    What it costs (tools/d2h_probe.py, one variant per process, 800x1333; 208 images/s with no
    result copy at all): masks as bits, 8 MB per image: 203 images/s with the bounded copy
    kernel (`pn_copy_stream`, 4 workgroups per field), 200-201 with 16 / 64 workgroups, 206
    with hipMemcpyAsync (a chip-wide blit KERNEL on this stack: rocprofv3 shows
    __amd_rocclr_copyBuffer, no SDMA copy record); masks as bytes, 51 MB per image: 193.
    Before the copies moved onto the producing chain stream they ran on a stream of this
    object's own, which HIP places on one of its 4 hardware queues by creation order --
    beside a stage-A stream or a chain stream: 182 or 196 images/s from one process to the
    next (bytes: 146 / 175 / 153 / 162 with 1 / 4 / 16 workgroups / hipMemcpyAsync; wide
    copies of 51 MB keep more PCIe writes in flight than the link drains and slow the
    concurrent GEMMs through the memory fabric they share).

    `pop()` waits for that batch's copies, checks its panoptic loops like `PSGTr.simple_test`
    (IndexError when every segment was filtered, pairnet_head.py:882) and returns Results
    whose arrays are VIEWS of the ring entry: valid until that entry is pushed again, i.e. for
    `ring` minus the number of batches still in flight more pushes -- ONE push when pop() is
    only called on a full ring; with `private_masks` the masks are private arrays
    (copy what must live longer).  The reference returns fresh arrays; this is the documented
    deviation that keeps allocation out of the loop."""

    PACK_MIN = 1 << 16     # bool fields of at least this many elements travel as bits

    def __init__(self, head, ring=4, copy_wgs=4, pack_masks=True, unpack_threads=4,
                 private_masks=False):
        """`copy_wgs`: workgroups of the device -> pinned-host copy kernel (`pn_copy_stream`):
        PCIe needs no width (one workgroup moves 7 GB/s; 4 keep a 51 MB image well under a
        step); 0 = hipMemcpyAsync.  `pack_masks`: bool fields (the 2R x H0 x W0 masks, 49 of
        the 51 MB) are packed to bits on the device (`pn_pack_bool_bits`), cross PCIe as
        6 MB and are expanded to the reference's numpy bool arrays by `unpack_threads` host
        threads in `pop()` (`pn_unpack_bits_host`)."""
        if head.device is None or head.device.type != "cuda":
            raise RuntimeError("ResultStreamer needs a head on an MI355X")
        self.head, self.device, self.ring = head, head.device, ring
        self.copy_wgs = max(0, int(copy_wgs))
        self.pack_masks, self.unpack_threads = bool(pack_masks), int(unpack_threads)
        # private_masks: the packed bool fields are expanded into arrays of a recycling pool
        # (`_BoolArrayPool`) instead of the ring entry: the Results' masks then belong to the
        # caller like the reference's, with no copy (needs pack_masks)
        self.mask_pool = _BoolArrayPool(ring + 2) if private_masks and pack_masks else None
        # the host half of the packed transfer runs on a worker thread of this object: it
        # waits for an entry's copy event and expands its bits while the caller's thread
        # keeps submitting images (both waits and the expansion run without the GIL)
        self._jobs, self._worker = queue.Queue(), None
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream()
        self.entries = [None] * ring      # dict(key, staging / host fields, event, jobs)
        self.head_i = self.tail_i = 0     # push / pop counters

    def _entry(self, key, results, n_jobs):
        """Ring entry for a batch whose fields have the shapes of `results`.  Every field owns
        FLAT buffers (device staging, pinned host, and for packed bool fields the host bool
        array) sized for the largest shape seen so far; the entry hands out views of their
        prefixes.  A keep-ratio evaluation set with many original image sizes therefore stops
        allocating once its largest image has passed (a multi-MB pinned or host allocation per
        image is an mmap / munmap pair against the busy GPU)."""
        idx = self.head_i % self.ring
        e = self.entries[idx]
        if e is not None and e["key"] == key:
            return e
        is_dev = lambda t: isinstance(t, torch.Tensor) and t.is_cuda
        packed = lambda t: (self.pack_masks and is_dev(t) and t.dtype == torch.bool
                            and t.numel() >= self.PACK_MIN)
        nbits = lambda t: ((t.numel() + 7) // 8 + 15) // 16 * 16
        # layout: per field 'p' (packed bits), 'd' (device tensor copied as is), 'h' (host const)
        layout = tuple(tuple("p" if packed(t) else "d" if is_dev(t) else "h" for t in tup)
                       for tup in results)
        if e is None or e["layout"] != layout or e["caps"]["jobs"] < n_jobs:
            e = self.entries[idx] = dict(layout=layout, caps=dict(jobs=max(1, n_jobs)), flat={},
                                         event=torch.cuda.Event(), ready=threading.Event(),
                                         error=None, jobs=0)
            e["dev_states"] = torch.empty(e["caps"]["jobs"], 16, dtype=torch.uint8, device=self.device)
            e["host_states"] = torch.empty(e["caps"]["jobs"], 16, dtype=torch.uint8, pin_memory=True)
        flat, caps = e["flat"], e["caps"]

        def grown(name, nbytes, make):
            """flat uint8 buffer `name` of at least nbytes (16-byte granules)."""
            nbytes = (max(int(nbytes), 1) + 15) // 16 * 16
            if caps.get(name, 0) < nbytes:
                flat[name] = make(nbytes)
                caps[name] = nbytes
            return flat[name]
        dev_u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=self.device)
        pin_u8 = lambda n: torch.empty(n, dtype=torch.uint8, pin_memory=True)
        host_b = lambda n: torch.empty(n, dtype=torch.bool)
        view = lambda buf, t: buf[:t.numel() * t.element_size()].view(t.dtype).view(t.shape)
        dev, host, bools = [], [], []
        for i, (tup, lay) in enumerate(zip(results, layout)):
            d_row, h_row, b_row = [], [], []
            for j, (t, kind) in enumerate(zip(tup, lay)):
                name = "%d.%d" % (i, j)
                if kind == "p":
                    nb = nbits(t)
                    d_row.append(grown("d" + name, nb, dev_u8)[:nb])
                    h_row.append(grown("h" + name, nb, pin_u8)[:nb])
                    b_row.append(grown("b" + name, t.numel(), host_b)[:t.numel()].view(t.shape))
                elif kind == "d":
                    nb = t.numel() * t.element_size()
                    d_row.append(view(grown("d" + name, nb, dev_u8), t))
                    h_row.append(view(grown("h" + name, nb, pin_u8), t))
                    b_row.append(None)
                else:                       # host-side constants pass through
                    d_row.append(None)
                    h_row.append(t)
                    b_row.append(None)
            dev.append(d_row)
            host.append(h_row)
            bools.append(b_row)
        e.update(key=key, dev=dev, host=host, bools=bools)
        return e

    def _to_host(self, src, dst):
        if self.copy_wgs > 0:
            hip.copy_stream(src, dst, self.copy_wgs)
        else:
            dst.copy_(src, non_blocking=True)

    @torch.no_grad()
    def push(self, results, pipe=None):
        """Stage one batch's `get_bboxes` tuples (ordered on the current stream) and queue
        their copies to the host.  `pipe`: the PipelinedHead they came from; its slot is
        released as soon as the staging copies are queued (`consumed`)."""
        if self.head_i - self.tail_i >= self.ring:
            raise RuntimeError("ResultStreamer ring is full: pop() before pushing more")
        jobs = tuple(getattr(results, "panoptic_jobs", ()))
        key = tuple(tuple((tuple(t.shape), t.dtype) if isinstance(t, torch.Tensor) and t.is_cuda
                          else None for t in tup) for tup in results) + (len(jobs),)
        e = self._entry(key, results, len(jobs))
        # (the ring entry's buffers are free again: pop() waited for its copies, and push()
        # refuses to overtake pop())
        # Results of a PipelinedHead are staged on the chain stream that produced them
        # (ordered behind get_bboxes there, not behind the image the caller has just queued on
        # its own stream), and their copies to the host follow on the same stream: a stream of
        # this object's own would share one of the 4 hardware queues with a pipeline stream
        # picked by creation order -- 182 or 196 images/s from one process to the next.
        src = getattr(results, "pipeline_stream", None) if pipe is not None else None
        cur = src if src is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(cur):
            for tup, stage, bools in zip(results, e["dev"], e["bools"]):
                for t, d, b in zip(tup, stage, bools):
                    if b is not None:
                        hip.pack_bool_bits(t.contiguous(), d)
                    elif d is not None:
                        d.copy_(t)
            for i, job in enumerate(jobs):
                e["dev_states"][i].copy_(job[0][:16])
            if pipe is not None:
                pipe.consumed(results, cur)
        copy_stream = cur if src is not None else self.stream
        if copy_stream is not cur:
            copy_stream.wait_stream(cur)
        with torch.cuda.stream(copy_stream):
            for stage, host in zip(e["dev"], e["host"]):
                for d, h in zip(stage, host):
                    if d is not None:
                        self._to_host(d, h)
            if jobs:
                self._to_host(e["dev_states"], e["host_states"])
            e["event"].record(copy_stream)
        e["jobs"] = len(jobs)
        self.head_i += 1
        if any(b is not None for bools in e["bools"] for b in bools):
            if self._worker is None:
                self._worker = threading.Thread(target=self._unpack_loop, daemon=True)
                self._worker.start()
            e["ready"].clear()
            self._jobs.put(e)
        else:
            e["ready"].set()

    def _unpack_loop(self):
        while True:
            e = self._jobs.get()
            if e is None:
                return
            try:
                e["event"].synchronize()
                for host, bools in zip(e["host"], e["bools"]):
                    for i, (h, b) in enumerate(zip(host, bools)):
                        if b is not None:
                            if self.mask_pool is not None:     # a private array per result
                                b = bools[i] = self.mask_pool.take(b.shape)
                            hip.unpack_bits_host(h, b, self.unpack_threads)
                e["error"] = None
            except BaseException as exc:      # handed to the thread that pops this entry
                e["error"] = exc
            e["ready"].set()

    def close(self):
        """Stop the worker thread (it is a daemon: optional)."""
        if self._worker is not None:
            self._jobs.put(None)
            self._worker.join()
            self._worker = None

    def pop(self):
        if self.tail_i >= self.head_i:
            raise RuntimeError("ResultStreamer is empty")
        e = self.entries[self.tail_i % self.ring]
        self.tail_i += 1
        e["ready"].wait()
        if e.get("error") is not None:
            raise e["error"]
        e["event"].synchronize()
        for i in range(e["jobs"]):
            nkeep, active, rounds, all_gone = e["host_states"][i].view(torch.int32).tolist()
            if all_gone:
                raise IndexError("every panoptic segment was filtered (the reference fails "
                                 "here too, pairnet_head.py:882)")
            if active:
                # The reference's drop-and-redo loop takes at most three rounds (LABNOTES.md
                # 1c) and hip.PAN_ROUNDS = 4 are enqueued, so this is unreachable unless that
                # bound is wrong; the slot's device buffers may already be reused, so the
                # loop cannot be continued here the way PSGTr.simple_test does -- fail loudly.
                raise RuntimeError("panoptic loop still active after %d rounds" % rounds)
        use_mask = self.head.use_mask
        out = []
        for host, bools in zip(e["host"], e["bools"]):
            fields = []
            for h, b in zip(host, bools):
                if b is not None and self.mask_pool is not None:
                    fields.append(self.mask_pool.hand_out(b))
                    continue
                h = b if b is not None else h          # (expanded by the worker thread)
                fields.append(h.numpy() if isinstance(h, torch.Tensor) else h)
            out.append(triplet2Result(tuple(fields), use_mask))
        return out

    def __len__(self):
        return self.head_i - self.tail_i


class PSGTr:
    """`PSGTr(SingleStageDetector)` of the reference, inference half."""

    def __init__(self, backbone, bbox_head, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None, neck=None):
        backbone = ConfigDict(backbone)
        # mmdet's ResNet returns the stages named by out_indices (default: all four)
        self.out_indices = tuple(backbone.get("out_indices", (0, 1, 2, 3)))
        self.neck = None
        if neck is not None:       # configs/deformable_detr/cross_r101_vg.py:20-29
            neck = dict(neck)
            if neck.pop("type", "ChannelMapper") != "ChannelMapper":
                raise NotImplementedError("necks built: ChannelMapper")
            self.neck = ChannelMapper(**neck)
        btype = backbone.get("type", "ResNet")
        if btype == "SwinTransformer":
            # pairnet_swinb.py:203-226; native only (swin.py)
            self.backbone = SwinTransformerHip(**{k: v for k, v in backbone.items() if k != "type"})
        elif btype == "ResNet" and backbone.get("depth", 50) in (50, 101):
            # the native fp32-MFMA backbone of backbone.py, channels_last features straight
            # into the head
            self.backbone = ResNet50Hip(depth=backbone.get("depth", 50))
        else:
            raise NotImplementedError("backbones built: ResNet depth 50 / 101 (pairnet.py, "
                                      "psgformer_r101_psg.py) and SwinTransformer "
                                      "(pairnet_swinb.py)")
        head_cfg = dict(bbox_head)
        heads = dict(CrossHead2=CrossHead2, CrossHeadBaseline=CrossHeadBaseline,
                     PSGTrHead2=PSGTrHead2, CrossHeadBBox=CrossHeadBBox)
        head_type = head_cfg.pop("type", "CrossHead2")
        if head_type not in heads:
            raise NotImplementedError("bbox_head.type must be one of %s" % sorted(heads))
        # (mmdet's SingleStageDetector hands the model-level train_cfg to the head -- the
        # base class psgtr.py:84-86 calls; here only
        # CrossHead2 reads it, for the forward values of its loss)
        self.bbox_head = heads[head_type](**head_cfg,
                                          train_cfg=train_cfg if head_type == "CrossHead2" else None,
                                          test_cfg=test_cfg or dict(max_per_img=100))
        self.num_classes = self.bbox_head.num_classes
        self.test_pipeline = None     # built by detect() (or set a preprocess.TestPipeline)
        self._mask_fetch = None       # bit-packed D2H of the masks into recycled arrays

    def to(self, device):
        self.__dict__.pop("_pipes", None)     # (cached pipelines hold the old device's streams)
        self.__dict__.pop("_calibrated", None)
        self.backbone.to(device)
        if self.neck is not None:
            self.neck.to(device)
        self.bbox_head.to(device)
        return self

    def eval(self):
        return self

    # ---- checkpoints: mmdet's layout `backbone.* / neck.* / bbox_head.*` ----
    def _parts(self):
        parts = [("backbone.", self.backbone), ("bbox_head.", self.bbox_head)]
        if self.neck is not None:
            parts.insert(1, ("neck.", self.neck))
        return parts

    def state_dict(self):
        out = OrderedDict()
        for prefix, mod in self._parts():
            for k, v in mod.state_dict().items():
                out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict=True):
        """The detector-level state dict of a reference checkpoint (keys prefixed with
        `backbone.`, `neck.`, `bbox_head.`; a leading `module.` of a DDP-saved file is dropped,
        as mmcv's load_checkpoint does).  Returns (missing, unexpected) key lists."""
        sd = {(k[len("module."):] if k.startswith("module.") else k): v
              for k, v in state_dict.items()}
        missing, unexpected, used = [], [], set()
        for prefix, mod in self._parts():
            sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
            used.update(prefix + k for k in sub)
            m, u = mod.load_state_dict(sub, strict=False)
            missing += [prefix + k for k in m]
            unexpected += [prefix + k for k in u]
        unexpected += [k for k in sd if k not in used]
        if strict and (missing or unexpected):
            raise RuntimeError("state_dict mismatch: missing %s unexpected %s"
                               % (missing[:5], unexpected[:5]))
        return missing, unexpected

    def extract_feat(self, img):
        """SingleStageDetector.extract_feat: backbone (the stages of `out_indices`) -> neck."""
        x = self.backbone(img)
        if len(x) == 4 and self.out_indices != (0, 1, 2, 3):
            x = tuple(x[i] for i in self.out_indices)
        return self.neck(x) if self.neck is not None else x

    @torch.no_grad()
    def simple_test(self, img, img_metas, rescale=False):
        """psgtr.py:148-156."""
        feat = self.extract_feat(img)
        results_list = self.bbox_head.simple_test(feat, img_metas, rescale=rescale)
        # the D2H point: the device-side panoptic loops are checked / finished here, where
        # the results are copied to the host anyway (raises like the reference when every
        # segment was filtered, pairnet_head.py:882)
        if hasattr(results_list, "panoptic_jobs"):
            self.bbox_head.panoptic_status(results_list)
        if self._mask_fetch is None:
            self._mask_fetch = _MaskFetcher()
        return [triplet2Result(t, self.bbox_head.use_mask, mask_fetch=self._mask_fetch)
                for t in results_list]

    @torch.no_grad()
    def val_losses(self, img, img_metas, gt_rels=None, gt_bboxes=None, gt_labels=None,
                   gt_masks=None, gt_bboxes_ignore=None, **kw):
        """The VALUES the reference's `PSGTr.forward_train` returns (psgtr.py:113-146), as a
        validation loss: extract_feat -> ground-truth masks zero-padded to the batch tensor's
        (H, W) and nearest-resized to (H // 2, W // 2) (:126-141, one kernel per image:
        `pn_gt_mask_prepare_u8`) -> `bbox_head.val_losses` (head forward + `loss`).  Forward
        only; the training iteration is `train_step`.

        `gt_masks[i]`: image i's instance masks [G, h, w], 0/1 -- a BitmapMasks-like object
        (`.to_ndarray()`), a numpy array or a tensor (host or device)."""
        if type(self.bbox_head) is not CrossHead2:     # (the siblings INHERIT val_losses)
            raise NotImplementedError("loss values are built for CrossHead2 only (%s has no "
                                      "loss forward here)" % type(self.bbox_head).__name__)
        x = self.extract_feat(img)
        gt_masks = self._prepare_gt_masks(img, gt_masks)
        return self.bbox_head.val_losses(x, img_metas, gt_rels, gt_bboxes, gt_labels, gt_masks,
                                         gt_bboxes_ignore, **kw)

    def _prepare_gt_masks(self, img, gt_masks):
        """PSGTr.forward_train's ground-truth mask preparation (psgtr.py:126-141): zero-pad to the
        batch tensor's (H, W), nearest-resize to (H // 2, W // 2); one kernel per image."""
        from . import hip
        if not getattr(self.bbox_head, "use_mask", True):
            return gt_masks
        assert gt_masks is not None
        H, W = int(img.shape[2]), int(img.shape[3])
        dev = self.bbox_head.device
        prepared = []
        with torch.cuda.device(dev):
            for each in gt_masks:
                m = each.to_ndarray() if hasattr(each, "to_ndarray") else each
                m = torch.as_tensor(m).to(dev)
                if m.dtype not in (torch.bool, torch.uint8):
                    m = (m != 0)
                if m.dim() != 3 or m.shape[1] > H or m.shape[2] > W:
                    raise ValueError("gt_masks: [G, h, w] with h <= %d, w <= %d, got %s"
                                     % (H, W, tuple(m.shape)))
                out = torch.empty((m.shape[0], H // 2, W // 2), dtype=torch.uint8, device=dev)
                if m.shape[0]:
                    hip.gt_mask_prepare(m.contiguous(), out, H, W)
                prepared.append(out)
        return prepared

    def trainer(self, train_backbone=True, **kw):
        """The per-iteration part of the reference's training (tools/train.py:115-241 with mmcv's
        runner: `forward_train` -> `losses.backward()` -> OptimizerHook(grad_clip) -> AdamW, under
        DDP) for this detector: a `pairnet_amd.TailTrainer` over every parameter the reference's R50
        config trains.  `train_backbone=False` (or a Swin backbone, which has no
        backward here) freezes the backbone; other keywords go to TailTrainer (lr, lr_mult, group,
        train_decoder=False / train_pixel_decoder=False for the frozen-detector regimes)."""
        from .backbone import ResNet50Hip
        from .train import TailTrainer
        if type(self.bbox_head) is not CrossHead2:
            raise NotImplementedError("training is built for CrossHead2 only")
        bb = self.backbone if (train_backbone and isinstance(self.backbone, ResNet50Hip)) else None
        kw.setdefault("train_decoder", True)
        kw.setdefault("train_pixel_decoder", True)
        if not (kw["train_decoder"] and kw["train_pixel_decoder"]):
            bb = None
        self._trainer = TailTrainer(self.bbox_head, backbone=bb, **kw)
        return self._trainer

    def train_step(self, img, img_metas=None, gt_rels=None, gt_bboxes=None, gt_labels=None,
                   gt_masks=None, point_coords=None):
        """One training iteration on one batch.  Two call forms:

        * `train_step(img, img_metas, gt_rels, gt_bboxes, gt_labels, gt_masks)` -- `forward_train`'s
          argument order (psgtr.py:113-146); returns the loss terms + `grad_norm` (device scalars);
        * `train_step(data_batch, optimizer)` -- mmdet's `BaseDetector.train_step`, what mmcv's
          `EpochBasedRunner.train` calls per iteration (`data_batch`: the collated dict with keys
          img / img_metas / gt_rels / gt_bboxes / gt_labels / gt_masks; `optimizer` is ignored:
          the step's own clip + AdamW have already run, so the runner needs NO OptimizerHook);
          returns mmdet's dict(loss, log_vars, num_samples) with `loss` = the sum of the terms whose
          name contains "loss" (`_parse_losses`).

        Ground-truth masks at image size are prepared like the reference's (pad to the batch
        tensor, nearest half-size), then `trainer().step` runs."""
        mmdet_form = isinstance(img, dict)
        if mmdet_form:
            data = img
            unwrap = lambda v: getattr(v, "data", v)        # (mmcv DataContainer)
            first = lambda v: v[0] if isinstance(v, (list, tuple)) and len(v) == 1 and \
                isinstance(v[0], (list, tuple)) else v
            img = unwrap(data["img"])
            if isinstance(img, (list, tuple)):
                img = img[0]
            img_metas = first(unwrap(data["img_metas"]))
            gt_rels, gt_labels = first(unwrap(data["gt_rels"])), first(unwrap(data["gt_labels"]))
            gt_masks = first(unwrap(data["gt_masks"]))
            gt_bboxes = first(unwrap(data.get("gt_bboxes")))
        tr = getattr(self, "_trainer", None) or self.trainer()
        gt_masks = self._prepare_gt_masks(img, gt_masks)
        x = img if tr.backbone is not None else self.extract_feat(img)
        out = tr.step(x, img_metas, gt_rels, gt_labels, gt_masks, point_coords=point_coords)
        if not mmdet_form:
            return out
        loss = sum(v for k, v in out.items() if "loss" in k)
        return dict(loss=loss, log_vars={k: float(v) for k, v in list(out.items()) + [("loss", loss)]},
                    num_samples=len(img_metas))

    @torch.no_grad()
    def detect(self, image, rescale=False):
        """Decoded uint8 (H, W, 3) BGR image (cv2 order, host or device) -> [Result]: the
        reference's test pipeline (configs/mask2former/pairnet.py:310-331) on the GPU
        (preprocess.TestPipeline), then `simple_test`."""
        img, metas = self._pipeline_of_images()(image)
        return self.simple_test(img, metas, rescale=rescale)

    def reserve(self, image_sizes, batch=1, depth=4, orig_sizes=()):
        """Size every buffer arena of the backbone and the head up front for [batch, 3, H, W]
        image tensors of the given `image_sizes` [(H, W)] (and post-processing at the original
        sizes `orig_sizes` [(H0, W0)]) on the `depth` slots of the pipeline.  Optional -- the
        arenas grow on demand, each growth costing one device wait (plans.py) -- and cheap: a
        loop that knows its envelope (Resize(img_scale=(1333, 800)): [(800, 1333), (1333, 800)])
        calls it once and never waits in flight.  Returns the bytes reserved, or None for a
        head that keeps per-shape plans (the box trunk)."""
        net, head = self.backbone, self.bbox_head
        if type(head)._plan is not CrossHead2._plan or self.neck is not None:
            return None
        piped = self._pipelines()
        a_slots = range(len(self.pipeline(depth).streams_a)) if piped else (0,)
        for H, W in image_sizes:
            net.reserve(batch, H, W, slots=a_slots)
            fs = net.feature_shapes(H, W)
            if len(fs) != 4:
                return None
            head.reserve(batch, [fs[3], fs[2], fs[1]], fs[0],
                         slots=range(depth) if piped else (0,), orig_sizes=orig_sizes)
        return net.arena_bytes() + head.arena_bytes()

    def pipeline(self, depth=4):
        """The detector's `PipelinedHead` for `depth` batches in flight, created once and kept:
        its four streams keep their hardware-queue placement (which `calibrate_pipeline`
        chooses empirically; worth ~8 % of the step, pipeline.py) across `stream()` /
        `stream_triplets()` / `dist.multi_gpu_test` calls."""
        from .pipeline import PipelinedHead
        pipes = self.__dict__.setdefault("_pipes", {})
        if depth not in pipes:
            saved = getattr(self.bbox_head, "grid_reserve", 0)
            pipes[depth] = PipelinedHead(self.bbox_head, depth=depth)
            self.bbox_head.grid_reserve = saved      # (set per run by _scheduled)
        return pipes[depth]

    @contextlib.contextmanager
    def _scheduled(self, depth):
        """The scheduling attributes a pipelined run sets on the head / backbone (hipGraph
        replay, the workgroup slots stage A leaves free), restored afterwards: grid_reserve is
        part of the stage graphs' key, so a later simple_test() finds its own graphs again."""
        head, net = self.bbox_head, self.backbone
        # both native backbones take a buffer slot per stage-A stream (two images' backbones
        # run side by side); only the ResNet replays as a hipGraph
        slots = isinstance(net, (ResNet50Hip, SwinTransformerHip))
        graphs = isinstance(net, ResNet50Hip)
        saved = (head.use_graphs, getattr(head, "grid_reserve", 0),
                 getattr(net, "use_graphs", None), getattr(net, "grid_reserve", 0))
        pipe = self.pipeline(depth)
        head.use_graphs, head.grid_reserve = True, pipe.grid_reserve
        if slots:
            net.grid_reserve = pipe.grid_reserve
        if graphs:
            net.use_graphs = True
        try:
            yield pipe, slots
        finally:
            head.use_graphs, head.grid_reserve = saved[0], saved[1]
            if slots:
                net.grid_reserve = saved[3]
            if graphs:
                net.use_graphs = saved[2]

    def _pipeline_of_images(self):
        if self.test_pipeline is None:
            from .config import test_pipeline_cfg
            from .preprocess import TestPipeline
            self.test_pipeline = TestPipeline.from_config(test_pipeline_cfg(),
                                                          device=self.bbox_head.device)
        return self.test_pipeline

    @staticmethod
    def _is_decoded(img):
        """A batch given as DECODED images: one uint8 (H, W, 3) array / tensor or a list of them
        (instead of the normalised float (B, 3, H, W) tensor `simple_test` takes)."""
        one = img[0] if isinstance(img, (list, tuple)) and len(img) else img
        return getattr(one, "dtype", None) in (torch.uint8, "uint8") or \
            str(getattr(one, "dtype", "")) == "uint8"

    def _submit(self, pipe, slots, img, metas, rescale):
        """Queue one batch: [test pipeline +] backbone + stage A on the pipeline's next
        stage-A stream."""
        head, net = self.bbox_head, self.backbone
        sl = pipe.count % len(pipe.streams_a)
        sa = pipe.streams_a[sl]
        sa.wait_stream(torch.cuda.current_stream(head.device))
        with torch.cuda.stream(sa):
            if self._is_decoded(img):
                # decoded uint8 BGR image(s): the reference's test pipeline (Resize keep-ratio,
                # Normalize, Pad, collate; configs/mask2former/pairnet.py:310-331) as one
                # kernel per image in front of the backbone, on this batch's stage-A stream,
                # into that stream's own grow-only buffer
                images = list(img) if isinstance(img, (list, tuple)) else [img]
                img, metas = self._pipeline_of_images().batch(images, slot=sl)
            if isinstance(img, torch.Tensor) and img.is_cuda:
                # a batch tensor the caller (or `dist.collate`) allocated on ITS stream is read
                # here on the stage-A stream: tell the caching allocator, or the block could be
                # handed to the next collate while the backbone of this batch still reads it
                # (the host runs several batches ahead of the device)
                img.record_stream(sa)
            feats = net(img, slot=sl) if slots else net(img)
            if len(feats) == 4 and self.out_indices != (0, 1, 2, 3):
                feats = tuple(feats[j] for j in self.out_indices)
            return pipe.submit(feats, metas, rescale=rescale)

    @torch.no_grad()
    def calibrate_pipeline(self, img, img_metas, depth=4, steps=8):
        """Choose the stream -> hardware-queue placement of this detector's pipeline on a
        representative batch (`PipelinedHead.calibrate` with the backbone in front, the way
        `stream()` queues it).  Optional, once per process; returns the per-rotation times."""
        self.warm_graphs(img, img_metas, depth=depth)
        with self._scheduled(depth) as (pipe, slots):
            times = pipe.calibrate(None, img_metas, steps=steps,
                                   submit=lambda: self._submit(pipe, slots, img, img_metas, False))
        self.__dict__.setdefault("_calibrated", set()).add(depth)
        return times

    @torch.no_grad()
    def warm_graphs(self, img, img_metas, depth=4, rescale=False):
        """Capture the hipGraphs of this batch's shape for every slot of the pipeline, at quiet
        points: graphs are only ever captured while no other stream of the process is executing
        (plans.quiet, LABNOTES R5.9), so a shape first met IN FLIGHT runs eagerly (within 0.5 %
        of the replay rate) until such a point comes.  This runs one batch at a time through
        each slot -- stage A, device wait, query chain + get_bboxes, device wait -- once eagerly
        and once capturing.  Optional; `calibrate_pipeline` starts with it."""
        if not self._pipelines():
            return
        head = self.bbox_head
        with self._scheduled(depth) as (pipe, slots):
            if pipe.queue:
                raise RuntimeError("warm_graphs() needs an empty pipeline")
            for _ in range(getattr(head, "graph_after", 1) + 1):
                for _ in range(depth):
                    self._submit(pipe, slots, img, img_metas, rescale)
                    torch.cuda.synchronize(head.device)
                    with torch.cuda.stream(pipe.streams_a[0]):
                        pipe.flush()
                    torch.cuda.synchronize(head.device)

    def pipeline_calibrated(self, depth=4):
        """Whether `calibrate_pipeline` has run for this depth (on the current device)."""
        return depth in self.__dict__.get("_calibrated", ())

    def _pipelined(self, batches, rescale, depth):
        """Generator behind `stream()` / `stream_triplets()`: queues every `(img, img_metas)`
        of `batches` -- backbone + stage A alternating between the pipeline's two stage-A
        streams, the query chains of older batches beside them -- and yields
        `(results, pipe)` for each batch, in order, `depth - 1` batches late.  `results`
        carries `pipeline_stream` (the chain stream `get_bboxes` ran on: queue reads THERE)
        and must be handed back with `pipe.consumed(results, stream)`."""
        with self._scheduled(depth) as (pipe, slots):
            if pipe.queue:
                raise RuntimeError("the detector's pipeline is in use by another generator")
            try:
                for img, metas in batches:
                    res = self._submit(pipe, slots, img, metas, rescale)
                    if res is not None:
                        yield res, pipe
                while pipe.queue:
                    with torch.cuda.stream(pipe.streams_a[0]):
                        res = pipe._finish(pipe.queue.pop(0))
                    yield res, pipe
            finally:
                if pipe.queue:          # (the consumer stopped early: finish what is queued)
                    with torch.cuda.stream(pipe.streams_a[0]):
                        pipe.flush()

    @classmethod
    def from_parts(cls, backbone, bbox_head, neck=None, out_indices=(0, 1, 2, 3)):
        """A detector around already-built parts (same attributes as the constructor sets)."""
        det = cls.__new__(cls)
        det.backbone, det.bbox_head, det.neck = backbone, bbox_head, neck
        det.out_indices = tuple(out_indices)
        det.num_classes = bbox_head.num_classes
        det.test_pipeline, det._mask_fetch = None, None
        return det

    def _pipelines(self):
        """Whether `_pipelined` can schedule this detector (the CrossHead2 family without a
        neck); other heads run one batch at a time."""
        return self.neck is None and bool(getattr(self.bbox_head, "use_mask", False))

    @torch.no_grad()
    def stream(self, batches, rescale=False, depth=4, ring=6, copy=True):
        """The throughput form of mmdet's `single_gpu_test` loop (tools/test.py:250-255):
        `batches` yields `(img, img_metas)` -- normalised (B, 3, H, W) device tensors, as
        `simple_test` takes them -- and this generator yields each batch's `[Result]`, in
        order, `depth - 1` batches late: backbone + stage A of consecutive batches alternate
        between two streams, the query chains of older batches run beside them, results reach
        the host through `ResultStreamer` (INTEGRATION.md 2b spelled out).  `copy=False`
        hands out views of the ring entry instead of private arrays: a yielded view is valid
        only until the NEXT batch is requested from this generator (the entry it lives in is
        the next one pushed); copy what must live longer.  Backbones other than the native
        ResNet run in front on the caller's stream."""
        head = self.bbox_head
        if not self._pipelines():
            for img, metas in batches:
                if self._is_decoded(img):
                    img, metas = self._pipeline_of_images().batch(
                        list(img) if isinstance(img, (list, tuple)) else [img])
                yield self.simple_test(img, metas, rescale=rescale)
            return
        out = ResultStreamer(head, ring=ring, private_masks=copy)
        own = (lambda rs: [self._own(r) for r in rs]) if copy else (lambda rs: rs)
        try:
            for res, pipe in self._pipelined(batches, rescale, depth):
                ready = own(out.pop()) if len(out) >= out.ring - 1 else None
                if ready is not None and not copy:
                    # a view: hand it out BEFORE its ring entry can be pushed again
                    yield ready
                    ready = None
                out.push(res, pipe)
                if ready is not None:
                    yield ready
            while len(out):
                yield own(out.pop())
        finally:
            out.close()

    @torch.no_grad()
    def stream_triplets(self, batches, rescale=False, depth=4):
        """The device-side form of `stream()` for the distributed test loop
        (`dist.multi_gpu_test`): yields a `dist.TripletBatch` per batch -- the `get_bboxes`
        tuples (device tensors), the query rows of their triplets, the chain stream they were
        produced on and a `release(stream)` hook -- without any D2H."""
        from .dist import TripletBatch
        head = self.bbox_head
        if not self._pipelines():
            for img, metas in batches:
                if self._is_decoded(img):
                    img, metas = self._pipeline_of_images().batch(
                        list(img) if isinstance(img, (list, tuple)) else [img])
                feat = self.extract_feat(img)
                res = head.simple_test(feat, metas, rescale=rescale)
                yield TripletBatch(res, *head.pair_positions(getattr(head, "_last_plan", None)))
            return
        for res, pipe in self._pipelined(batches, rescale, depth):
            sub, obj = head.pair_positions(head._last_plan)
            yield TripletBatch(res, sub, obj, stream=res.pipeline_stream,
                               release=lambda s, r=res, p=pipe: p.consumed(r, s))

    @staticmethod
    def _own(r):
        """A Result whose arrays no longer alias the streamer's ring."""
        import numpy as np
        # (the masks already are a private array of the streamer's pool: `private_masks`)
        q = Result(**{k: (v.copy() if isinstance(v, np.ndarray) and k != "masks" else v)
                      for k, v in r.__dict__.items()})
        if isinstance(q.formatted_masks, dict):
            q.formatted_masks = dict(pan_results=q.pan_results)
        return q

    def forward(self, img=None, img_metas=None, return_loss=False, rescale=False, **kw):
        """mmdet's `model(return_loss=False, rescale=True, img=[..], img_metas=[..])`."""
        if return_loss:
            raise NotImplementedError("model(return_loss=True) + autograd is not how this detector "
                                      "trains: `val_losses` gives forward_train's loss VALUES, "
                                      "`pairnet_amd.TailTrainer(det.bbox_head).step(det.extract_feat(img), ...)` "
                                      "(backbone=det.backbone to train it too) runs one iteration")
        if isinstance(img, (list, tuple)):
            img, img_metas = img[0], img_metas[0]
        return self.simple_test(img, img_metas, rescale=rescale)

    __call__ = forward


def load_checkpoint(model, filename, map_location="cpu", strict=False, trust=False):
    """`mmcv.runner.load_checkpoint(model, filename, map_location="cpu")` (tools/test.py:240)
    for local files: a torch-saved dict with `state_dict` (and `meta`: CLASSES, PREDICATES,
    ...), or a bare state dict.  Loads into `model` (a PSGTr, or any head / backbone / neck of
    this package) and returns the checkpoint dict, like mmcv.

    The file is read with `weights_only=True` (tensors, containers, strings, numbers: what a
    state dict plus `meta` needs); a checkpoint that pickles other objects is refused unless
    the caller vouches for it with `trust=True` (mmcv unpickles anything -- arbitrary code)."""
    import pickle
    try:
        ckpt = torch.load(filename, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        # only what the restricted unpickler raises for a non-tensor payload: I/O errors
        # (missing file, permissions) propagate unchanged, and so does a corrupt archive
        # (torch raises RuntimeError for that too: told apart by its message)
        refused = isinstance(e, pickle.UnpicklingError) or "weights_only" in str(e).lower() \
            or "unsupported global" in str(e).lower()
        if not refused:
            raise
        if not trust:
            raise RuntimeError(
                "%s holds pickled objects beyond tensors / plain containers (%s); pass "
                "trust=True to unpickle it anyway (executes code from the file)" % (filename, e))
        ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    if missing or unexpected:
        import warnings
        warnings.warn("load_checkpoint: %d missing key(s) (first: %s), %d unexpected (first: %s)"
                      % (len(missing), missing[:2], len(unexpected), unexpected[:2]))
    return ckpt if isinstance(ckpt, dict) and "state_dict" in ckpt else dict(state_dict=sd, meta={})


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """`build_detector(cfg.model)` (tools/test.py:235-236)."""
    cfg = dict(cfg)
    if cfg.pop("type", "PSGTr") != "PSGTr":
        raise NotImplementedError("only type='PSGTr'")
    model_train_cfg = cfg.pop("train_cfg", None)
    return PSGTr(cfg["backbone"], cfg["bbox_head"],
                 train_cfg=model_train_cfg if model_train_cfg is not None else train_cfg,
                 test_cfg=cfg.get("test_cfg", test_cfg), neck=cfg.get("neck"))
