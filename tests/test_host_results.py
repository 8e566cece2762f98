"""CPU tests of the host side of the serving path: the library's COCO RLE encoder under concurrent callers (persistent helper
pool, round 4). The encoder itself is pinned against the reference format in tests/test_host_logic.py."""


def test_rle_encoder_concurrent_callers_share_the_pool():
    """Round 4: the RLE helpers are a persistent pool inside the library; the serving loop calls the encoder from several Python
    threads at once (`RleCollector` encodes `depth` batches concurrently). Concurrent calls with different inputs and thread counts
    must return exactly what a single-threaded call returns."""
    import concurrent.futures as cf
    import numpy as np
    from cgg_amd import ops
    rng = np.random.default_rng(5)
    jobs = []
    for i in range(12):
        n, H, W = int(rng.integers(1, 40)), int(rng.choice([64, 96, 200])), int(rng.choice([64, 128, 136]))
        m = np.zeros((n, H, W), dtype=bool)
        for k in range(n):                       # a few rectangles + salt noise: short and long runs
            for _ in range(3):
                y0, x0 = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
                m[k, y0:y0 + int(rng.integers(4, H - y0)), x0:x0 + int(rng.integers(4, W - x0))] ^= True
            m[k] ^= rng.random((H, W)) < 0.01
        bits = np.packbits(m, axis=-1, bitorder='little')
        jobs.append((bits, W, int(rng.choice([1, 2, 7, 32, 64]))))
    want = [ops.rle_encode_bitmasks(b, w, threads=1) for b, w, _ in jobs]
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        for _ in range(3):
            got = list(ex.map(lambda j: ops.rle_encode_bitmasks(j[0], j[1], threads=j[2]), jobs))
            assert got == want
