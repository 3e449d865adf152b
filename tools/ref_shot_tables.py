#!/usr/bin/env python3
"""Developer tool (GPU): which sub-sample table did the binary behind the README renderings use?  (finding 3)
With ONE fixed table in every render the per-pixel z-test is decided on edges, where the value is set by which cells of the pixel the
samples fall in: the right table brings the reference's |z| > 6 fraction down to the control's; any other leaves it 100x - 10000x above.
Candidate tables are built here in Python (ChaCha with 8 / 12 / 20 rounds behind rand's seed_from_u64 + Fisher-Yates shuffle) for several
cell-size rules, and passed to the library as `sample_xy` (scaled to the grid the kernels divide by).
usage: python tools/ref_shot_tables.py [shot ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustray_amd import capi  # noqa: E402
from rustray_amd.flat import make_config  # noqa: E402
from tests.helpers import camera_for  # noqa: E402
from tests.test_ref_shots import SHOTS, _cell_size_of, _z_stats, box2, load_shot, scene_of_2022  # noqa: E402

M32 = 0xffffffff


def rotl(v, n):
    return ((v << n) | (v >> (32 - n))) & M32


class ChaCha:
    """rand_chacha's ChaChaXRng behind SeedableRng::seed_from_u64 (PCG32 expansion), 32-bit output stream."""
    def __init__(self, seed, rounds):
        state, self.key = seed, []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & 0xffffffffffffffff
            xs = (((state >> 18) ^ state) >> 27) & M32
            rot = state >> 59
            self.key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & M32)
        self.rounds, self.counter, self.block, self.pos = rounds, 0, [], 16

    def refill(self):
        x = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + self.key + [self.counter & M32, (self.counter >> 32) & M32, 0, 0]
        w = list(x)

        def qr(a, b, c, d):
            w[a] = (w[a] + w[b]) & M32; w[d] = rotl(w[d] ^ w[a], 16)
            w[c] = (w[c] + w[d]) & M32; w[b] = rotl(w[b] ^ w[c], 12)
            w[a] = (w[a] + w[b]) & M32; w[d] = rotl(w[d] ^ w[a], 8)
            w[c] = (w[c] + w[d]) & M32; w[b] = rotl(w[b] ^ w[c], 7)
        for _ in range(self.rounds // 2):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        self.block = [(a + b) & M32 for a, b in zip(w, x)]
        self.counter += 1
        self.pos = 0

    def next_u32(self):
        if self.pos >= 16:
            self.refill()
        v = self.block[self.pos]
        self.pos += 1
        return v

    def below(self, rng_):
        zone = ((rng_ << (32 - rng_.bit_length())) - 1) & M32
        while True:
            m = self.next_u32() * rng_
            if (m & M32) <= zone:
                return m >> 32


def shuffled_cells(cs, n, rounds=12, seed=0, transpose=False):
    cells = [(x, y) for x in range(cs) for y in range(cs)]
    if transpose:
        cells = [(y, x) for x, y in cells]
    g = ChaCha(seed, rounds)
    for i in range(len(cells) - 1, 0, -1):
        j = g.below(i + 1)
        cells[i], cells[j] = cells[j], cells[i]
    return np.asarray(cells[:n], np.int64)


def next_pow2(v):
    p = 1
    while p < v:
        p <<= 1
    return p


def candidates(spp):
    cs = _cell_size_of(spp)   # what the kernels divide by
    out = {}

    def put(name, cells, cs_c):
        out[name] = np.ascontiguousarray(np.clip(np.round(cells * (cs / cs_c)), 0, cs - 1).astype(np.uint16))
    put("HEAD rule, ChaCha12 (the built-in table)", shuffled_cells(cs, spp), cs)
    put("HEAD rule, ChaCha12, (y, x) order", shuffled_cells(cs, spp, transpose=True), cs)
    put("HEAD rule, ChaCha20", shuffled_cells(cs, spp, rounds=20), cs)
    put("HEAD rule, ChaCha8", shuffled_cells(cs, spp, rounds=8), cs)
    h = max(next_pow2(spp) // 2, 1)
    put("cell_size = next_pow2(samples) / 2, ChaCha12", shuffled_cells(h, spp), h)
    put("cell_size = next_pow2(samples) / 2, ChaCha20", shuffled_cells(h, spp, rounds=20), h)
    q = int(np.ceil(np.sqrt(spp)))
    put("cell_size = ceil(sqrt(samples)), ChaCha12", shuffled_cells(q, spp), q)
    put("cell_size = ceil(sqrt(samples)), unshuffled", np.asarray([(x, y) for x in range(q) for y in range(q)][:spp], np.int64), q)
    return out


if __name__ == "__main__" and "--edges" not in sys.argv:
    K = 6
    built_in = capi.sample_table(32)[0]
    assert np.array_equal(np.asarray(built_in, np.int64), shuffled_cells(_cell_size_of(32), 32)), "the Python ChaCha12 shuffle is not the library's"
    for name in (sys.argv[1:] or ["floor_monkey", "room_kbert"]):
        ref, mask, meta = load_shot(name)
        fs = scene_of_2022(name)
        cam = camera_for(fs, 1280, 720).c_struct()
        r = ref.astype(np.float64)
        keep = np.ones(ref.shape, bool)
        spp = meta["samples"]
        print(f"{name} ({spp} spp): |z| > 6 fraction of the reference | of the control, one fixed table in all {K} + 1 renders")
        with capi.DeviceScene(fs, 0) as ds:
            ds.set_compat(1)
            for label, table in candidates(spp).items():
                half = np.stack([box2(ds.render(cam, make_config(samples=spp, monte_carlo=True, seed=1000 + i), aux=False, sample_xy=table)["rgba"][..., :3]).astype(np.float32)
                                 for i in range(K + 1)])
                mu, sd = half[:K].mean(axis=0).astype(np.float64), half[:K].std(axis=0, ddof=1).astype(np.float64)
                a, c = _z_stats(r, mu, sd, keep, K), _z_stats(half[K].astype(np.float64), mu, sd, keep, K)
                print(f"  {label:52s} {a['gt6']:.2e} | {c['gt6']:.2e}   (|z| > 4: {a['gt4']:.2e} | {c['gt4']:.2e}, bias {a['bias']:+.3f})", flush=True)


def edge_correlation():
    """Second question: ONE unknown table for all pixels, or a table drawn per pixel?  Along an edge the residual (value - mean over random
    tables) of neighbouring pixels is strongly correlated when the whole frame shares one table (the coverage is a smooth function of
    the sub-pixel phase) and independent when every pixel draws its own.  Compared: the reference; single renders of ours (one table per
    frame); and composites that take every pixel from a randomly chosen one of K renders with different tables (a table per pixel)."""
    K = 12
    for name in ("floor_monkey", "room_kbert", "room_spheres"):
        ref, mask, meta = load_shot(name)
        fs = scene_of_2022(name)
        cam = camera_for(fs, 1280, 720).c_struct()
        spp = meta["samples"]
        cs = _cell_size_of(spp)
        cells = np.stack(np.meshgrid(np.arange(cs), np.arange(cs), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.uint16)
        rng = np.random.default_rng(4242)
        with capi.DeviceScene(fs, 0) as ds:
            ds.set_compat(1)
            half = np.stack([box2(ds.render(cam, make_config(samples=spp, monte_carlo=True, seed=3000 + i), aux=False,
                                            sample_xy=np.ascontiguousarray(cells[rng.permutation(len(cells))[:spp]]))["rgba"][..., :3]).astype(np.float64)
                             for i in range(K)])
        mu, sd = half.mean(axis=0), half.std(axis=0, ddof=1)
        # edge pixels: where the sampling pattern matters (sd over tables well above the quantisation), green channel
        ch = 1
        edge = sd[..., ch] > 1.0

        def corr(img):
            d = (img[..., ch] - mu[..., ch]) / np.maximum(sd[..., ch], 1e-9)
            pair = edge[:, :-1] & edge[:, 1:]
            a, b = d[:, :-1][pair], d[:, 1:][pair]
            pv = edge[:-1, :] & edge[1:, :]
            a2, b2 = d[:-1, :][pv], d[1:, :][pv]
            return float(np.corrcoef(a, b)[0, 1]), float(np.corrcoef(a2, b2)[0, 1]), int(pair.sum() + pv.sum())
        singles = [corr(half[i]) for i in range(3)]
        pick = rng.integers(0, K, ref.shape[:2])
        comp = [corr(np.take_along_axis(half, rng.integers(0, K, ref.shape[:2])[None, ..., None], axis=0)[0]) for _ in range(3)]
        r = corr(ref.astype(np.float64))
        print(f"{name}: lag-1 correlation of the edge residuals (horizontal, vertical neighbours; {r[2]} pairs)")
        print(f"  reference                                  {r[0]:+.3f} {r[1]:+.3f}")
        print("  our single renders (one table per frame)   " + "  ".join(f"{a:+.3f} {b:+.3f}" for a, b, _ in singles))
        print("  our composites (a table per pixel)         " + "  ".join(f"{a:+.3f} {b:+.3f}" for a, b, _ in comp), flush=True)


if __name__ == "__main__" and "--edges" in sys.argv:
    edge_correlation()
