"""Device results -> host results for serving / evaluation.

The reference hands every mask to the host as its own (H, W) bool array (open_set/models/maskformer.py:205-208: ~300
`.cpu().numpy()` calls and 315 MB per 1024^2 image) and the evaluation path then run-length encodes each of them
(pycocotools `mask.encode` inside the dataset's `results2json`). `RleCollector` produces that END format directly:

  device   masks stay bit-packed ((n, H, W/8) uint8 from `cgg_instance_masks_picks(bitpack=1)`: 39 MB per image)
  copy     ONE asynchronous device->host copy per tensor into pinned staging buffers on a side stream, ordered after the
           producing stream by an event -- the GPU pipeline is never blocked by a pageable `.cpu()`
  host     `cgg_rle_encode_bitmasks` (C++ host threads of the extension) -> COCO RLE dicts, in a worker thread that
           overlaps the next batch's device work (ctypes releases the GIL)

Result per image: {eval_type: (bbox_results, segm_results)} with the reference's per-class list layout
(`bbox2result` arrays; `segm_results[label]` = list of {'size': [H, W], 'counts': bytes}), i.e. what
`results2json` would build from the reference's arrays.
"""
import concurrent.futures as cf

import numpy as np
import torch

from . import ops


class RleCollector:

    def __init__(self, device, num_classes, depth=3, rle_threads=None):
        """num_classes: {eval_type: number of classes} (the fusion head's all / novel / base class counts)."""
        self.device = torch.device(device)
        self.num_classes = dict(num_classes)
        self.depth = depth
        import os
        from . import runtime
        # the encoder's cost is per RUN; `depth` batches are encoded concurrently, each on this many host threads. The budget is
        # the CPUs the process may really use (cgroup quota, not `nproc`): oversubscribing it gets every thread throttled
        self.rle_threads = rle_threads or int(os.environ.get('CGG_RLE_THREADS', 0)) or \
            max(2, min(16, (runtime.effective_cpu_count() - 2) // max(depth, 1)))
        self.copy_stream = torch.cuda.Stream(self.device)
        self.pool = cf.ThreadPoolExecutor(max_workers=depth)
        self._slots = [dict(buffers={}, event=None, future=None) for _ in range(depth)]
        self._n = 0

    def _pinned(self, slot, key, like):
        buf = slot['buffers'].get(key)
        if buf is None or buf.shape != like.shape or buf.dtype != like.dtype:
            buf = torch.empty(like.shape, dtype=like.dtype, pin_memory=True)
            slot['buffers'][key] = buf
        return buf

    def submit(self, results, width=None):
        """results: list (one per image) of {eval_type: (labels, bboxes (n, 5), masks (n, H, W/8) uint8)} DEVICE tensors
        (`simple_test(..., device_results=True, mask_bits=True)`). Returns a future -> list of host result dicts.
        The device tensors may be overwritten once `wait_copied(ticket)` returned (ticket = the future's `.ticket`)."""
        slot = self._slots[self._n % self.depth]
        if slot['future'] is not None:
            slot['future'].result()                       # the slot's previous batch has left its pinned buffers
        self._n += 1
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        staged = []
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            for i, res in enumerate(results):
                per = {}
                for key, val in res.items():
                    if not (isinstance(val, (tuple, list)) and len(val) == 3 and torch.is_tensor(val[2])):
                        continue
                    labels, bboxes, masks = val
                    packed = masks.dtype != torch.bool
                    if not packed:                      # shapes outside the bit-packing kernel: packed on the host below
                        masks = masks.view(torch.uint8)
                    w_px = int(width) if width is not None else masks.shape[-1] * (8 if packed else 1)
                    hl = self._pinned(slot, (i, key, 'l'), labels)
                    hb = self._pinned(slot, (i, key, 'b'), bboxes)
                    hm = self._pinned(slot, (i, key, 'm'), masks)
                    hl.copy_(labels, non_blocking=True)
                    hb.copy_(bboxes, non_blocking=True)
                    hm.copy_(masks, non_blocking=True)
                    per[key] = (hl, hb, hm, w_px, packed)
                staged.append(per)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        slot['event'] = done
        fut = self.pool.submit(self._encode, staged, done)
        fut.copied = done
        slot['future'] = fut
        return fut

    @staticmethod
    def wait_copied(fut):
        """Host-side wait until the batch's device tensors have been copied out (they may then be overwritten)."""
        fut.copied.synchronize()

    def _encode(self, staged, done):
        done.synchronize()
        out = []
        for per in staged:
            res = {}
            for key, (hl, hb, hm, W, packed) in per.items():
                labels = hl.numpy().astype(np.int64)
                bboxes = hb.numpy().copy()
                bits = hm if packed else np.packbits(hm.numpy(), axis=-1, bitorder='little')
                rles = ops.rle_encode_bitmasks(bits, W, threads=self.rle_threads)
                ncls = int(self.num_classes[key])
                bbox_results = [bboxes[labels == c, :] for c in range(ncls)]
                segm = [[] for _ in range(ncls)]
                for lab, r in zip(labels.tolist(), rles):
                    segm[lab].append(r)
                res[key] = (bbox_results, segm)
            out.append(res)
        return out

    def close(self):
        self.pool.shutdown(wait=True)


def fusion_class_counts(fusion_head):
    return dict(all_results=getattr(fusion_head, 'all_classes', None), novel_results=getattr(fusion_head, 'novel_classes', None),
                base_results=getattr(fusion_head, 'base_classes', None))


def rle_to_mask(rle):
    """Decoder of the COCO counts string (tests / consumers without pycocotools): -> (H, W) bool."""
    H, W = rle['size']
    s = rle['counts']
    counts, i, p = [], 0, 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if i > 2:
            x += counts[i - 2]
        counts.append(x)
        i += 1
    flat = np.zeros(H * W, dtype=bool)
    pos, v = 0, False
    for c in counts:
        if v:
            flat[pos:pos + c] = True
        pos += c
        v = not v
    return flat.reshape((W, H)).T
