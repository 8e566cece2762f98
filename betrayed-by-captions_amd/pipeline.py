"""Software pipeline across batches for inference throughput on one MI355X.

One forward of the detector is a chain of stages with very different hardware profiles:

  * `backbone`      -- convolutions / GEMMs; the stride-16/32 tail is many small launches that cannot fill 256 CUs;
  * `head._encode`  -- MSDeformAttn pixel decoder, K/V projections, packed mask feature: big GEMMs and gathers
                       (with the backbone: 4.1 of the 6.4 ms step at configs[1]);
  * `head._decode`+ post-processing -- the 9-layer query decoder: ~300 dependent launches on M = B*Q = 200 rows,
                       7-56 workgroups each -- latency-bound, the chip is almost idle (2.3 ms).

Back to back they serialise. Here every stage runs on its own HIP stream and batch k's stage i overlaps batch k+1's
stage i-1: the latency-bound chains hide under the throughput-bound ones (the hardware schedules workgroups of all
queues concurrently). Each stage is captured ONCE per buffer slot into a hipGraph (`slots` copies, so that a stage never
overwrites what the next stage of an older batch is still reading); a step is then one graph launch and one or two
event edges per stage, no Python in between. Per-batch latency is unchanged -- this is a throughput device, exactly
like double buffering a data loader.

Correctness: the pipelined results are the sequential results (tests/test_head_gpu.py::test_stage_pipeline).
"""
import torch


def _record_stream(obj, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


class StagePipeline:
    """`submit(x)` enqueues one batch and returns its slot; `wait(slot)` makes the caller's stream wait for that
    batch's results (`results[slot]`); `flush()` waits for everything.

    stages   -- list of callables; stage 0 takes the (static copy of the) input batch, stage i the output of stage
                i-1; the last stage's return value is the result. They are called under `torch.no_grad()`.
    example  -- an example input batch (shape / dtype / device are frozen into the graphs)
    """

    def __init__(self, stages, example, slots=None, warmup=2, priorities=None, streams=None, tail=None):
        """tail -- optional callable applied EAGERLY (not captured) to the last graph stage's output: post-processing with
                   data-dependent host reads (the panoptic fusion head reads segment counts: no hipGraph can hold that). It runs
                   on its own stream ONE BATCH BEHIND the graph stages -- `submit(k + 1)` first enqueues batch k + 1's graphs, then
                   runs batch k's tail -- so the host waits of the tail overlap the next batch's device work."""
        self.stages = list(stages)
        n = len(self.stages)
        self.tail = tail
        self.slots = slots if slots is not None else max(2, n)
        if tail is not None and self.slots < 2:
            # the eager tail runs ONE batch behind the graphs: with a single slot batch k + 1's graphs would overwrite the stage
            # outputs batch k's tail has not read yet
            raise ValueError('StagePipeline: a tail needs slots >= 2')
        dev = example.device
        self.dev = dev
        import os
        if priorities is None and os.environ.get('CGG_PIPE_PRIO'):
            priorities = [int(v) for v in os.environ['CGG_PIPE_PRIO'].split(',')]
        self.streams = list(streams) if streams is not None else \
            [torch.cuda.Stream(dev, priority=(priorities[i] if priorities else 0)) for i in range(n)]
        self.inputs = [torch.empty_like(example) for _ in range(self.slots)]
        self.done = [[torch.cuda.Event() for _ in range(self.slots)] for _ in range(n)]
        self.graphs = [[None] * self.slots for _ in range(n)]
        self.outs = [[None] * self.slots for _ in range(n)]
        self._n = 0
        cur = torch.cuda.current_stream(dev)
        for s in range(self.slots):
            self.inputs[s].copy_(example)
        # every tensor the model's caches hand out while the stages warm up and are captured (positional encodings, projection
        # tables, folded biases of THIS image shape) is kept for the pipeline's lifetime: the graphs replay raw addresses, and the
        # caches are bounded / replace their entry when another shape arrives
        from . import runtime as _rt
        self._scope = _rt.keepalive_scope()
        self._keepalive = self._scope.__enter__()
        try:
            self._capture(cur, dev, warmup)
        finally:
            self._scope.__exit__(None, None, None)
        self.results = self.outs[-1]
        self._tail_pending = None
        if tail is not None:
            self._init_tail(tail, dev)

    def _capture(self, cur, dev, warmup):
        with torch.no_grad():
            # eager warm-up on the stage streams: solver searches, weight packing, allocator pools settle before capture
            for _ in range(max(warmup, 1)):
                x = self.inputs[0]
                prev = cur
                for i, f in enumerate(self.stages):
                    self.streams[i].wait_stream(prev)
                    with torch.cuda.stream(self.streams[i]):
                        x = f(x)
                    prev = self.streams[i]
                cur.wait_stream(prev)
            torch.cuda.synchronize(dev)
            for s in range(self.slots):
                x = self.inputs[s]
                for i, f in enumerate(self.stages):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.streams[i]):
                        x = f(x)
                    self.graphs[i][s] = g
                    self.outs[i][s] = x
                    torch.cuda.synchronize(dev)

    def _init_tail(self, tail, dev):
        self.tail_stream = torch.cuda.Stream(dev)
        self.tail_done = [torch.cuda.Event() for _ in range(self.slots)]
        self.results = [None] * self.slots
        self._tail_ran = [False] * self.slots
        with torch.no_grad():                                 # warm-up (allocator, first-call setup) on the tail stream
            self.tail_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.tail_stream):
                tail(self.outs[-1][0])
            torch.cuda.synchronize(dev)

    def _run_tail(self, slot):
        st = self.tail_stream
        with torch.no_grad(), torch.cuda.stream(st):
            st.wait_event(self.done[-1][slot])
            self.results[slot] = self.tail(self.outs[-1][slot])
            self.tail_done[slot].record(st)
        self._tail_ran[slot] = True

    def submit(self, x):
        s = self._n % self.slots
        self._n += 1
        n = len(self.stages)
        cur = torch.cuda.current_stream(self.dev)
        self.streams[0].wait_stream(cur)                          # `x` was produced on the caller's stream
        for i in range(n):
            st = self.streams[i]
            with torch.cuda.stream(st):
                if i > 0:
                    st.wait_event(self.done[i - 1][s])            # this batch's previous stage
                if i + 1 < n:
                    st.wait_event(self.done[i + 1][s])            # the slot's previous batch has left the next stage
                else:
                    st.wait_event(self.done[i][s])
                    if self.tail is not None and self._tail_ran[s]:
                        st.wait_event(self.tail_done[s])          # the slot's previous batch has left the eager tail
                if i == 0 and x is not self.inputs[s]:
                    self.inputs[s].copy_(x, non_blocking=True)
                self.graphs[i][s].replay()
                self.done[i][s].record(st)
        if self.tail is not None:
            prev, self._tail_pending = self._tail_pending, s
            if prev is not None:
                self._run_tail(prev)                              # one batch behind: its host reads overlap this batch's graphs
        return s

    def wait(self, slot):
        """Block the CALLER'S stream (not the host) until `results[slot]` is complete; returns the results."""
        if self.tail is not None:
            if self._tail_pending == slot:
                self._run_tail(slot)
                self._tail_pending = None
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(self.tail_done[slot])
            # the tail's results were allocated from the TAIL stream's pool: tell the allocator the caller's stream reads them too,
            # or their blocks could be handed to later tail-stream work while e.g. an asynchronous D2H copy on the caller's stream
            # is still reading them
            _record_stream(self.results[slot], cur)
            return self.results[slot]
        torch.cuda.current_stream(self.dev).wait_event(self.done[-1][slot])
        return self.results[slot]

    def flush(self):
        """Make the caller's stream wait for everything submitted so far."""
        cur = torch.cuda.current_stream(self.dev)
        if self.tail is not None and self._tail_pending is not None:
            self._run_tail(self._tail_pending)
            self._tail_pending = None
        for st in self.streams + ([self.tail_stream] if self.tail is not None else []):
            cur.wait_stream(st)


def detector_pipeline(model, example, metas, stages=3, defer_tail=True, **decode_kwargs):
    """Pipeline of a `MaskFormerOpen` detector: 2 stages = (backbone + head encode | decode + post-processing),
    3 stages = (backbone | head encode | decode + post-processing). `defer_tail` moves the K / V projections and the
    mask-feature packing from the encode stage (the longer one) to the head of the decode stage."""
    head = model.panoptic_head
    if getattr(model.panoptic_fusion_head, 'panoptic_mode', False):
        # panoptic post-processing reads data-dependent sizes on the host (kept queries, segment areas): the graph stages end with
        # the query decoder, the fusion head runs eagerly one batch behind (`StagePipeline(tail=...)`)
        post = lambda out: model.stage_post(out, metas, **decode_kwargs)      # noqa: E731
        if stages in (2, 4):
            fns = [lambda x: model.stage_encode(x, defer_tail=defer_tail), lambda enc: model.stage_head(enc, metas, **decode_kwargs)]
        else:
            fns = [model.extract_feat, lambda f: head._encode(f, defer_tail=defer_tail),
                   lambda enc: model.stage_head(enc, metas, **decode_kwargs)]
        return StagePipeline(fns, example, tail=post)
    if stages == 2:
        fns = [lambda x: model.stage_encode(x, defer_tail=defer_tail),
               lambda enc: model.stage_decode(enc, metas, **decode_kwargs)]
    elif stages == 3:
        fns = [model.extract_feat, lambda f: head._encode(f, defer_tail=defer_tail),
               lambda enc: model.stage_decode(enc, metas, **decode_kwargs)]
    elif stages == 4:      # THREE functions: encode | query decoder | post-processing (device results only); `len(pipe.stages)` is
                           # what callers report -- the argument names the 2-stage split "+ a post-processing stage"
        fns = [lambda x: model.stage_encode(x, defer_tail=defer_tail),
               lambda enc: model.stage_head(enc, metas, **decode_kwargs),
               lambda out: model.stage_post(out, metas, **decode_kwargs)]
    elif stages == 5:      # FOUR functions: backbone | pixel decoder | query decoder | post-processing (device results only)
        fns = [model.extract_feat, lambda f: head._encode(f, defer_tail=defer_tail),
               lambda enc: model.stage_head(enc, metas, **decode_kwargs),
               lambda out: model.stage_post(out, metas, **decode_kwargs)]
    else:
        raise ValueError('stages must be 2, 3, 4 or 5')
    return StagePipeline(fns, example)


def TwoStagePipeline(model, example, metas, **decode_kwargs):
    return detector_pipeline(model, example, metas, stages=2, **decode_kwargs)
