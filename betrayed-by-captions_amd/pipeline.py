"""Two-stage software pipeline for inference throughput on one MI355X.

One forward of the detector has two very different halves:

  * `stage_encode`  -- backbone, MSDeformAttn pixel decoder, K/V projections, packed mask feature: big GEMMs,
    convolutions and gathers that fill the chip (4.1 of the 6.4 ms step at configs[1]);
  * `stage_decode`  -- the 9-layer query decoder + mask logits + post-processing: ~300 dependent launches on
    M = B*Q = 200 rows, 7-56 workgroups each -- latency-bound, the 256 CUs are almost idle (2.3 ms).

Back to back they serialise. Here batch k's decode runs on one HIP stream while batch k+1's encode runs on another:
the latency-bound chain hides under the throughput-bound one (the hardware schedules workgroups of both queues
concurrently). Each stage is captured ONCE per buffer slot into a hipGraph (two slots, so that encode(k+1) never
overwrites what decode(k) is still reading); a step is then two graph launches and two event edges, no Python in
between. Per-batch latency is unchanged -- this is a throughput device, exactly like double buffering a data loader.

Correctness: the pipelined results are the sequential results (tests/test_head_gpu.py::test_two_stage_pipeline).
"""
import torch


class TwoStagePipeline:
    """`submit(img)` enqueues one batch; results come back in order from `submit` (the batch submitted `depth - 1`
    calls earlier, None while the pipeline fills) and from `flush()`.

    model      -- a detector with `stage_encode(img)` / `stage_decode(enc, metas, **kw)` (detectors.MaskFormerOpen)
    example    -- an example input batch (shape / dtype / device are frozen into the graphs)
    metas      -- img_metas of every batch (fixed geometry)
    """

    def __init__(self, model, example, metas, slots=2, warmup=2, **decode_kwargs):
        self.model = model
        self.metas = metas
        self.kw = decode_kwargs
        self.slots = slots
        dev = example.device
        self.s_enc = torch.cuda.Stream(dev)
        self.s_dec = torch.cuda.Stream(dev)
        self.inputs = [torch.empty_like(example) for _ in range(slots)]
        self.enc_done = [torch.cuda.Event() for _ in range(slots)]
        self.dec_done = [torch.cuda.Event() for _ in range(slots)]
        self.g_enc, self.g_dec, self.enc_out, self.results = [], [], [], []
        self._n = 0
        cur = torch.cuda.current_stream(dev)
        # eager warm-up on the side streams: solver searches, weight packing and allocator pools settle before capture
        for s in range(slots):
            self.inputs[s].copy_(example)
        self.s_enc.wait_stream(cur)
        with torch.no_grad():
            for _ in range(max(warmup, 1)):
                with torch.cuda.stream(self.s_enc):
                    enc = model.stage_encode(self.inputs[0])
                self.s_dec.wait_stream(self.s_enc)
                with torch.cuda.stream(self.s_dec):
                    model.stage_decode(enc, metas, **self.kw)
                self.s_enc.wait_stream(self.s_dec)
            torch.cuda.synchronize(dev)
            for s in range(slots):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.s_enc):
                    enc = model.stage_encode(self.inputs[s])
                self.g_enc.append(g)
                self.enc_out.append(enc)
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.s_dec):
                    res = model.stage_decode(enc, metas, **self.kw)
                self.g_dec.append(g)
                self.results.append(res)
                torch.cuda.synchronize(dev)

    def submit(self, img):
        """Enqueue one batch (device tensor, copied into the slot's static input on the encode stream). Returns the
        slot index whose `results[slot]` will hold this batch's output once `dec_done[slot]` has fired."""
        s = self._n % self.slots
        self._n += 1
        cur = torch.cuda.current_stream(img.device)
        self.s_enc.wait_stream(cur)                       # `img` was produced on the caller's stream
        with torch.cuda.stream(self.s_enc):
            self.s_enc.wait_event(self.dec_done[s])       # the previous user of this slot has been fully decoded
            if img is not self.inputs[s]:
                self.inputs[s].copy_(img, non_blocking=True)
            self.g_enc[s].replay()
            self.enc_done[s].record(self.s_enc)
        with torch.cuda.stream(self.s_dec):
            self.s_dec.wait_event(self.enc_done[s])
            self.g_dec[s].replay()
            self.dec_done[s].record(self.s_dec)
        return s

    def wait(self, slot):
        """Block the CALLER'S stream (not the host) until `results[slot]` is complete; returns the results."""
        torch.cuda.current_stream().wait_event(self.dec_done[slot])
        return self.results[slot]

    def flush(self):
        """Make the caller's stream wait for everything submitted so far."""
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.s_enc)
        cur.wait_stream(self.s_dec)
