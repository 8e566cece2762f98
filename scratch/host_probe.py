"""Where the host-results path spends its time: per-batch durations of the RLE C call and the Python around it."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from cgg_amd import ops, runtime, synthetic, host_results
from cgg_amd.host_results import RleCollector, fusion_class_counts
from cgg_amd.pipeline import detector_pipeline

class A: pass
args = A(); args.size = 1024; args.batch = 2; args.queries = 100; args.precision = 'fp32'
dev = torch.device('cuda', 0)
runtime.set_precision('fp32')
cfg, model = bench.build_model(args, dev)
img = torch.randn(2, 3, 1024, 1024, device=dev)
metas = synthetic.img_metas(2, 1024, 1024)
stats = []
orig = ops.rle_encode_bitmasks
def timed_rle(bits, width, threads=8):
    arr = bits.numpy() if torch.is_tensor(bits) else bits
    t0 = time.perf_counter(); _ = int(arr.reshape(-1).view(np.uint64).sum()); stats.append(('read13MB', time.perf_counter() - t0, 0))
    cp = arr.copy()
    t0 = time.perf_counter(); r2 = orig(cp, width, threads); stats.append(('rle_on_copy', time.perf_counter() - t0, 0))
    t0 = time.perf_counter(); r = orig(bits, width, threads); stats.append(('rle', time.perf_counter() - t0, threading.get_ident())); return r
ops.rle_encode_bitmasks = timed_rle
orig_enc = RleCollector._encode
def timed_enc(self, staged, done):
    t0 = time.perf_counter(); done.synchronize(); t1 = time.perf_counter(); r = orig_enc(self, staged, done)
    stats.append(('encode_total', time.perf_counter() - t1, 0)); stats.append(('wait_copy', t1 - t0, 0)); return r
RleCollector._encode = timed_enc
with runtime.precision_scope('fp32'):
    pipe = detector_pipeline(model, img, metas, stages=3, defer_tail=0, rescale=True, device_results=True, mask_bits=True)
    col = RleCollector(dev, fusion_class_counts(model.panoptic_fusion_head))
    def run(n):
        depth = 2
        futs, in_copy, pending, copied = [], [], [], {}
        for _ in range(n):
            while len(in_copy) > depth + 1:
                RleCollector.wait_copied(in_copy.pop(0))
            ev = copied.pop(pipe._n % pipe.slots, None)
            if ev is not None:
                pipe.streams[-1].wait_event(ev)
            pending.append(pipe.submit(img))
            while len(pending) > depth:
                old = pending.pop(0)
                f = col.submit(pipe.wait(old)); futs.append(f); in_copy.append(f); copied[old] = f.copied
        for old in pending:
            futs.append(col.submit(pipe.wait(old)))
        return [r for f in futs for r in f.result()]
    run(3); torch.cuda.synchronize()
    print('affinity main:', len(os.sched_getaffinity(0)), 'of', os.cpu_count(), 'OMP', os.environ.get('OMP_NUM_THREADS'), 'torch threads', torch.get_num_threads(), flush=True)
    def cgstat():
        d = {}
        try:
            for l in open('/sys/fs/cgroup/cpu.stat'):
                k, v = l.split(); d[k] = int(v)
        except OSError:
            pass
        return d
    def task_cpu():
        out = {}
        for t in os.listdir('/proc/self/task'):
            try:
                f = open(f'/proc/self/task/{t}/stat').read()
                name = f[f.index('(') + 1:f.rindex(')')]
                rest = f[f.rindex(')') + 2:].split()
                out[t] = (name, (int(rest[11]) + int(rest[12])) / os.sysconf('SC_CLK_TCK'))
            except Exception:
                pass
        return out
    for rep in range(2):
        c0, k0 = cgstat(), task_cpu()
        stats.clear()
        t0 = time.perf_counter(); res = run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        by = {}
        for k, v, _ in stats: by.setdefault(k, []).append(v)
        c1, k1 = cgstat(), task_cpu()
        busy = sorted(((k1[t][1] - k0.get(t, (None, 0))[1], k1[t][0]) for t in k1), reverse=True)[:6]
        print(f'   wall {dt:.2f} s; cgroup usage {(c1.get("usage_usec", 0) - c0.get("usage_usec", 0)) / 1e6:.2f} cpu-s, throttled {(c1.get("throttled_usec", 0) - c0.get("throttled_usec", 0)) / 1e6:.2f} s in {c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0)} periods; threads {len(k1)}; busiest:', [(round(a, 2), n) for a, n in busy], flush=True)
        print(f'rep {rep}: {len(res) / dt:.1f} img/s;', '; '.join(f'{k}: n={len(v)} mean {1e3 * sum(v) / len(v):.2f} ms max {1e3 * max(v):.2f}' for k, v in by.items()), flush=True)
